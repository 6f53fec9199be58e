cd $GRAFT_REPO_ROOT
for v in 7 3 5 0 7; do
  PB_WGRAD_STREAM=$v python bench.py --no-cpu-baseline --no-probe --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('PB_WGRAD_STREAM=$v', round(r['ms_per_step'],2), 'ms/step')"
done
for t in 128 256 384; do
  PB_WG_TARGET=$t python bench.py --no-cpu-baseline --no-probe --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('PB_WG_TARGET=$t', round(r['ms_per_step'],2), 'ms/step')"
done
