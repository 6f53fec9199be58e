#!/bin/bash
# Same-box A/B of the split-bf16 step between environment settings:  gpurun -- 'bash tools/x3_ab.sh 16 "PB_X3_PLANES=0" "PB_X3_PLANES=1"'
B=$1; shift
run() { env "$@" python bench.py --precision bf16x3 --batch $B --steps 5 --warmup 2 --no-probe --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('B=$B $*', round(r['ms_per_step'],2), r['train_loss'])"; }
for r in 1 2; do for arm in "$@"; do run $arm; done; done
