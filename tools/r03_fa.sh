#!/bin/bash
# attention forward: 4 waves x 32 queries (PB_FA_FWD8=0) vs 8 waves x 16 queries (default), same box, alternating
mkdir -p gpurun_out/r03
for r in 1 2; do
  echo "== fwd4"; PB_FA_FWD8=0 python tools/flash_bench.py
  echo "== fwd8"; python tools/flash_bench.py
done 2>&1 | tee gpurun_out/r03/fa_fwd8_ab.txt
timeout 1200 python -m pytest tests/test_bench_shape_gpu.py tests/test_kernels_gpu.py tests/test_packed_gpu.py -q -m gpu -x 2>&1 | tail -3
for r in 1 2; do
  echo "== step fwd4"; PB_FA_FWD8=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
  echo "== step fwd8"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
