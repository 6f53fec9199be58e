#!/bin/bash
# The rocprofv3 evidence of a round, in one go (run on the GPU box: gpurun -- 'bash tools/profile_round.sh r03'); every summary lands in
# gpurun_out/<tag>prof/ under the name it is committed with in profiles/:
#   <tag>_bench_b32_packed_kernel_stats_{two_streams,one_stream}.txt   kernel trace of bench.py (shipped schedule / one stream)
#   <tag>_gemm_fc1_pmc_M<rows>.json                                     FETCH_SIZE / WRITE_SIZE passes of the dominant kernel at the step's row counts
#   <tag>_sq_counters_attention.txt                                    SQ counters of the head_dim-64 attention kernels (4 separate --pmc passes)
#   <tag>_decode_kernel_stats.txt                                      kernel trace of the KV-cached decode (graph replay per token)
#   <tag>_step_byte_budget.txt                                         FETCH_SIZE x 2 + WRITE_SIZE per kernel family over one step (two --pmc passes over bench.py)
#   <tag>_bench_default.json                                           the driver-style bench line of the same box
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${TAG}prof; mkdir -p $O; rm -rf $O/prof_* $O/pmc_* $O/p[1-4]
python bench.py > $O/${TAG}_bench_default.json 2> $O/bench_default.err; tail -c 400 $O/${TAG}_bench_default.json; echo
ROWS=$(python - <<PY
import json
d=json.loads([l for l in open('$O/${TAG}_bench_default.json').read().splitlines() if l.startswith('{')][-1])
print(d['rows']['encoder_side'], d['rows']['decoder_side'])
PY
)
export PB_GEMM_LDS_TAG=1          # the traced launches tell their GEMM problem through group_segment_size (tools/rocpd_stats.py splits the gemm rows by it)
rocprofv3 --kernel-trace -d $R/$O/prof_2s -- python3 bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 2 > $O/prof_2s.log 2>&1
DB=$(ls $O/prof_2s/*/*.db | head -1)
{ echo "# rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 2   (6 steps in the file; packed rows, shipped two-stream schedule)"; python tools/rocpd_overlap.py $DB | sed 's/^/# /'; python tools/rocpd_stats.py $DB 50; } > $O/${TAG}_bench_b32_packed_kernel_stats_two_streams.txt
PB_WGRAD_STREAM=0 rocprofv3 --kernel-trace -d $R/$O/prof_1s -- python3 bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 2 > $O/prof_1s.log 2>&1
DB=$(ls $O/prof_1s/*/*.db | head -1)
{ echo "# PB_WGRAD_STREAM=0 rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 2   (6 steps in the file; packed rows, one stream: undisturbed per-kernel durations)"; python tools/rocpd_overlap.py $DB | sed 's/^/# /'; python tools/rocpd_stats.py $DB 50; } > $O/${TAG}_bench_b32_packed_kernel_stats_one_stream.txt
rm -rf $O/prof_2s $O/prof_1s
unset PB_GEMM_LDS_TAG
# whole-step byte budget: FETCH_SIZE and WRITE_SIZE in separate passes over the same bench command (4 steps in each trace)
MS=$(python - <<PY
import json
d=json.loads([l for l in open('$O/${TAG}_bench_default.json').read().splitlines() if l.startswith('{')][-1])
print(d['ms_per_step'])
PY
)
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/bud_f -- python3 bench.py --no-cpu-baseline --no-probe --steps 2 --warmup 2 > $O/bud_f.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/bud_w -- python3 bench.py --no-cpu-baseline --no-probe --steps 2 --warmup 2 > $O/bud_w.log 2>&1
{ echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu-baseline --no-probe --steps 2 --warmup 2; tools/pmc_step_budget.py"; python tools/pmc_step_budget.py $O/bud_f $O/bud_w $MS; } > $O/${TAG}_step_byte_budget.txt 2>&1
rm -rf $O/bud_f $O/bud_w
for M in $ROWS; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_f$M -- python3 tools/pmc_gemm.py --M=$M > $O/pmc_f$M.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_w$M -- python3 tools/pmc_gemm.py --M=$M > $O/pmc_w$M.log 2>&1
  python tools/pmc_to_json.py $O/pmc_f$M $O/pmc_w$M $M 3072 768 $O/${TAG}_gemm_fc1_pmc_M$M.json
  rm -rf $O/pmc_f$M $O/pmc_w$M
done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/p$i -- python3 tools/pmc_flash.py > $O/p$i.log 2>&1
done
python tools/sq_summary.py $O/p1 $O/p2 $O/p3 $O/p4 > $O/${TAG}_sq_counters_attention.txt 2>&1; rm -rf $O/p[1-4]
timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof_dec -- python3 bench.py --mode decode --no-cpu-baseline --steps 200 > $O/prof_dec.log 2>&1
{ echo "# rocprofv3 --kernel-trace -- python3 bench.py --mode decode --no-cpu-baseline --steps 200   (12L/768d, S = 1024, B = 1; device-sampled decode: 8 tokens per hipGraph replay, 75 launches per token)"; python tools/rocpd_decode.py $(ls $O/prof_dec/*/*.db | head -1); tail -1 $O/prof_dec.log | cut -c1-400; } > $O/${TAG}_decode_kernel_stats.txt 2>&1
rm -rf $O/prof_dec
ls -la $O; head -12 $O/${TAG}_bench_b32_packed_kernel_stats_two_streams.txt
