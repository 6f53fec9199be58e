#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_shape_gpu.py tests/test_packed_gpu.py -x -q -m gpu 2>&1 | tail -5
for cfg in "1 0" "1 1" "0 1"; do
  set -- $cfg
  PB_PACK_ROWS=$1 PB_GEMM_TAIL=$2 timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/bench_p$1_t$2.json 2> gpurun_out/bench_p$1_t$2.err
done
python - <<'PY'
import json
for n in ('p1_t0', 'p1_t1', 'p0_t1'):
    j = json.loads(open('gpurun_out/bench_%s.json' % n).read().strip().splitlines()[-1])
    print(n, 'ms/step', round(j['ms_per_step'], 2), 'median', round(j['ms_per_step_median_hip_events'], 2), 'gemm in step', round(j['roofline']['families_in_step']['gemm']['ms_per_step'], 2))
PY
