#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_packed_gpu.py tests/test_parallel_gpu.py tests/test_bench_shape_gpu.py -x -q -m gpu 2>&1 | tail -3
for r in 0 1 0 1 0 1; do
  PB_WGRAD_BATCH=$r timeout 600 python bench.py --no-cpu-baseline --no-probe --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PB_WGRAD_BATCH=$r', round(j['ms_per_step'],2), round(j['ms_per_step_median_hip_events'],2), j['train_loss'])"
done
