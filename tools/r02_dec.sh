cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -s -k "decode or generate or g8" -p no:cacheprovider 2>&1 | tail -8
for v in 0 1; do PB_DECODE_SPLIT=$v timeout 600 python bench.py --mode decode --steps 512 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('split=$v', round(r['ms_per_step'],4), 'ms/token')"; done
rm -rf $O/prof_dec2; timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof_dec2 -- python3 bench.py --mode decode --no-cpu-baseline --steps 200 > $O/prof_dec2.log 2>&1
python tools/rocpd_decode.py $(ls $O/prof_dec2/*/*.db | head -1) | tee $O/decode_stats2.txt
