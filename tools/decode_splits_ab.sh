#!/bin/bash
# Same-box A/B of the decode attention split counts (workgroups per head of the self- / cross-attention launches): bench.py --mode decode, 400 tokens.
#   gpurun -- 'bash tools/decode_splits_ab.sh'
run() { env PB_DECODE_SPLITS_SELF=$1 PB_DECODE_SPLITS_CROSS=$2 python bench.py --mode decode --steps 400 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('self $1 cross $2:', round(r['ms_per_step'],4), 'ms/token, rewinds', r['decode_info'].get('rewinds'))"; }
for r in 1 2; do
  run 16 16; run 8 8; run 16 8; run 8 16; run 4 8; run 4 4; run 2 4
done
