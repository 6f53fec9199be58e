#!/bin/bash
# Same-box A/B of the pre-train step between environment settings, alternating, two rounds (boxes differ by +-3 %: never compare
# across gpurun calls). Each argument is one arm: a space-separated list of VAR=value (use PB_X=0 for "the defaults").
#   gpurun -- 'bash tools/ab_env.sh "PB_X=0" "PB_NO_PACK_PREFETCH=1" "PB_GEMM_FLAGS=65536 PB_NO_PACK_PREFETCH=1"'
run() { env "$@" python bench.py --no-cpu-baseline --no-probe --steps 20 --warmup 8 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$*', round(r['ms_per_step'],2), round(r['ms_per_step_median_hip_events'],2), r['train_loss'])"; }
for r in 1 2; do
  for arm in "$@"; do run $arm; done
done
