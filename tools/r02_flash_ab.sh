# same-box A/B of the attention forward (PB_FLASH_PP=0: 4-wave kernel, 1: 8-wave ping-pong) + the flash tests
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k flash -x -p no:cacheprovider 2>&1 | tail -4
timeout 600 python -m pytest tests/test_bench_shape_gpu.py -m gpu -q -k "flash" -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -6
for r in 1 2; do
  echo "== 4-wave"; PB_FLASH_PP=0 python tools/flash_bench.py
  echo "== ping-pong"; python tools/flash_bench.py
done
