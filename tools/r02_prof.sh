# round 2 profiles: kernel-trace of bench.py (two-stream default and one-stream), per-kernel stats + stream overlap
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r02; mkdir -p $O
rm -rf $O/prof_2s $O/prof_1s
rocprofv3 --kernel-trace -d $R/$O/prof_2s -- python3 bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 2 > $O/prof_2s.log 2>&1
DB=$(ls $O/prof_2s/*/*.db | head -1)
{ echo "# rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 2   (6 steps in the file; shipped two-stream schedule)"; python tools/rocpd_overlap.py $DB | sed 's/^/# /'; python tools/rocpd_stats.py $DB 45; } > $O/r02_bench_b32_kernel_stats_two_streams.txt
PB_WGRAD_STREAM=0 rocprofv3 --kernel-trace -d $R/$O/prof_1s -- python3 bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 2 > $O/prof_1s.log 2>&1
DB=$(ls $O/prof_1s/*/*.db | head -1)
{ echo "# PB_WGRAD_STREAM=0 rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 2   (6 steps in the file; everything on one stream: undisturbed per-kernel durations)"; python tools/rocpd_overlap.py $DB | sed 's/^/# /'; python tools/rocpd_stats.py $DB 45; } > $O/r02_bench_b32_kernel_stats_one_stream.txt
rm -rf $O/prof_2s/*/*.db $O/prof_1s/*/*.db
head -12 $O/r02_bench_b32_kernel_stats_one_stream.txt; tail -1 $O/prof_2s.log | cut -c1-300
