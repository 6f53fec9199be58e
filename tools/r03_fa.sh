#!/bin/bash
# attention kernels: old library (ab/old.so) vs the working tree, same box, alternating
for r in 1 2; do
  echo "== old"; PB_LIB_PATH=$PWD/ab/old.so python tools/flash_bench.py
  echo "== new"; python tools/flash_bench.py
done
timeout 900 python -m pytest tests/test_bench_shape_gpu.py tests/test_kernels_gpu.py tests/test_packed_gpu.py -q -m gpu -x -k "flash or attention" 2>&1 | tail -3
