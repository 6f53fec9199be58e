#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_packed_gpu.py -x -q -m gpu -k transpose 2>&1 | tail -2
for r in 0 1; do
  PB_DGRAD_NT=$r timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PB_DGRAD_NT=$r', round(j['ms_per_step'],2), j['roofline']['families_in_step']['gemm'])
for r in j['roofline']['top_ops_in_step']:
    if 'gemm' in r['op']: print('   %-70s %5.1f %8.1f us %s' % (r['op'], r['calls_per_step'], r['avg_us'], round(r['tflops'])))"
done
