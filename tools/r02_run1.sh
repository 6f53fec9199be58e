# round 2, GPU call 1: the whole -m gpu suite (new bench-shape / replay / reducer tests included), the default bench line, kernel-trace profiles
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -q -rfE -s -p no:cacheprovider > gpurun_out/r02/t1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02/t1.log
tail -5 gpurun_out/r02/t1.log
timeout 600 python bench.py > gpurun_out/r02/bench1.json 2> gpurun_out/r02/bench1.err
tail -c 3000 gpurun_out/r02/bench1.json
