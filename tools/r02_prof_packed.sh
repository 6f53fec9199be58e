#!/bin/bash
# round 2, packed step: kernel-trace stats of bench.py (shipped two-stream schedule and one stream) + PMC traffic of fc1 at the packed row counts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r02p; mkdir -p $O
rm -rf $O/prof_2s $O/prof_1s $O/pmc_*
rocprofv3 --kernel-trace -d $R/$O/prof_2s -- python3 bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 2 > $O/prof_2s.log 2>&1
DB=$(ls $O/prof_2s/*/*.db | head -1)
{ echo "# rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 2   (6 steps in the file; packed rows, shipped two-stream schedule)"; python tools/rocpd_overlap.py $DB | sed 's/^/# /'; python tools/rocpd_stats.py $DB 50; } > $O/r02_bench_b32_packed_kernel_stats_two_streams.txt
PB_WGRAD_STREAM=0 rocprofv3 --kernel-trace -d $R/$O/prof_1s -- python3 bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 2 > $O/prof_1s.log 2>&1
DB=$(ls $O/prof_1s/*/*.db | head -1)
{ echo "# PB_WGRAD_STREAM=0 rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 2   (6 steps in the file; packed rows, one stream: undisturbed per-kernel durations)"; python tools/rocpd_overlap.py $DB | sed 's/^/# /'; python tools/rocpd_stats.py $DB 50; } > $O/r02_bench_b32_packed_kernel_stats_one_stream.txt
rm -rf $O/prof_2s/*/*.db $O/prof_1s/*/*.db
for M in 26624 26880; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_f$M -- python3 tools/pmc_gemm.py --M=$M > $O/pmc_f$M.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_w$M -- python3 tools/pmc_gemm.py --M=$M > $O/pmc_w$M.log 2>&1
  python tools/pmc_to_json.py $O/pmc_f$M $O/pmc_w$M $M 3072 768 $O/r02_gemm_fc1_pmc_M$M.json
done
head -14 $O/r02_bench_b32_packed_kernel_stats_one_stream.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
