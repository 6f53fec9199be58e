#!/bin/bash
# round 3: same-box A/B of the step (alternating runs): row split off (PB_GEMM_FLAGS=65536), count prefetch off, default
O=gpurun_out/r03; mkdir -p $O
run() { env "$@" python bench.py --no-cpu-baseline --no-probe --steps 20 --warmup 8 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$*', round(r['ms_per_step'],2), round(r['ms_per_step_median_hip_events'],2))"; }
for r in 1 2; do
  run PB_X=0
  run PB_GEMM_FLAGS=65536
  run PB_NO_PACK_PREFETCH=1
  run PB_GEMM_FLAGS=65536 PB_NO_PACK_PREFETCH=1
done
