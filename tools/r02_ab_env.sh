#!/bin/bash
# Same-box A/B of one environment toggle on the whole training step (round 2 used it for PB_RING, PB_DGRAD_NT, PB_EVENT_MODE,
# PB_SUB_LAST, PB_SIDE_TAIL, PB_FWD_GEMM_FLAGS, PB_WG_TARGET, PB_NO_PIPELINE_UPDATES ...): alternating runs of bench.py, one line each.
#   gpurun -- 'bash tools/r02_ab_env.sh PB_RING 1 2 1 2'
var=$1; shift
for v in "$@"; do
  env $var=$v timeout 600 python bench.py --no-cpu-baseline --no-probe --steps 40 --warmup 10 2>/dev/null | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', 'ms/step', round(j['ms_per_step'],2), 'median', round(j['ms_per_step_median_hip_events'],2), 'loss', j['train_loss'])"
done
