#!/bin/bash
# round 3: GEMM tests (race screen: three passes) + same-box A/B of the step for PB_GEMM_FLAGS variants
mkdir -p gpurun_out/r03; O=gpurun_out/r03
for i in 1 2 3; do timeout 1200 python -m pytest tests/test_bench_shape_gpu.py tests/test_packed_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -k "gemm or nt_ or dgrad or wgrad or fc1 or logits or tail or row_split" > $O/gemm_tests_$i.log 2>&1; tail -2 $O/gemm_tests_$i.log; done
run() { env "$@" python bench.py --no-cpu-baseline --no-probe --steps 20 --warmup 8 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$*', round(r['ms_per_step'],2), round(r['ms_per_step_median_hip_events'],2), r['train_loss'])"; }
for r in 1 2 3; do
  run PB_X=0
  run PB_GEMM_FLAGS=131072
  run PB_GEMM_FLAGS=196608
done
PB_WGRAD_STREAM=0 python bench.py --no-cpu-baseline --no-probe --steps 10 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('one stream default', round(r['ms_per_step'],2))"
PB_WGRAD_STREAM=0 PB_GEMM_FLAGS=196608 python bench.py --no-cpu-baseline --no-probe --steps 10 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('one stream no-stream no-rowsplit', round(r['ms_per_step'],2))"
