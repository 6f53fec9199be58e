run() { env "$@" python bench.py --force-reducer --no-cpu-baseline --no-probe --steps 20 --warmup 8 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$*', round(r['ms_per_step'],2), r['train_loss'])"; }
for r in 1 2; do
python bench.py --no-cpu-baseline --no-probe --steps 20 --warmup 8 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('plain', round(r['ms_per_step'],2), r['train_loss'])"
run PB_DP_RESERVE_CUS=16 PB_DP_GRADS=f32
run PB_DP_RESERVE_CUS=8 PB_DP_GRADS=f32
run PB_DP_RESERVE_CUS=0 PB_DP_GRADS=f32
run PB_DP_RESERVE_CUS=16 PB_DP_GRADS=bf16
run PB_DP_RESERVE_CUS=0 PB_DP_GRADS=bf16
done
