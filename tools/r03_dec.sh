#!/bin/bash
# round 3: decode at the configs[3] shape -- graph replay vs direct launches vs the round-2 per-launch loop, then kernel traces
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r03; mkdir -p $O
for g in 1 0 -1; do PB_DECODE_GRAPH=$g timeout 600 python bench.py --mode decode --steps 200 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('graph=$g', round(r['ms_per_step'],4), 'ms/token', r.get('decode_info'))"; done
for g in 0 1; do
rm -rf $O/prof_dec_g$g; PB_DECODE_GRAPH=$g timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof_dec_g$g -- python3 bench.py --mode decode --no-cpu-baseline --steps 200 > $O/prof_dec_g$g.log 2>&1
python tools/rocpd_decode.py $(ls $O/prof_dec_g$g/*/*.db | head -1) > $O/decode_stats_g$g.txt 2>&1; head -40 $O/decode_stats_g$g.txt
rm -rf $O/prof_dec_g$g
done
