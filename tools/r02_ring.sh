#!/bin/bash
mkdir -p gpurun_out
for r in 0 1 -1 0 1; do
  PB_SIDE_PRIORITY=$r PB_DEBUG=1 timeout 600 python bench.py --no-cpu-baseline --no-probe --steps 40 --warmup 10 2>/dev/null | grep "^{\|side stream" | python -c "
import json,sys
L=sys.stdin.read().strip().splitlines()
j=json.loads(L[-1]); print('PB_SIDE_PRIORITY=$r', round(j['ms_per_step'],2), round(j['ms_per_step_median_hip_events'],2), j['train_loss'], [l for l in L if 'side stream' in l][:1])"
done
