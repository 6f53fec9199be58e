# same-box A/B of the attention kernels (ab/r02_base.so = library before the change) + the flash tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k flash -x -p no:cacheprovider 2>&1 | tail -5
timeout 600 python -m pytest tests/test_bench_shape_gpu.py -m gpu -q -k "flash or step_matches_fp32_step_cfg2 or g10" -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -12
for r in 1 2; do
  echo "== base"; PB_LIB_PATH=$PWD/ab/r02_base.so python tools/flash_bench.py
  echo "== new"; python tools/flash_bench.py
done
