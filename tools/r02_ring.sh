#!/bin/bash
mkdir -p gpurun_out
for r in 0 1 0 1 0 1; do
  PB_SIDE_TAIL=$r timeout 600 python bench.py --no-cpu-baseline --no-probe --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PB_SIDE_TAIL=$r', round(j['ms_per_step'],2), round(j['ms_per_step_median_hip_events'],2), j['train_loss'])"
done
