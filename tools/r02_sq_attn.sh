#!/bin/bash
# SQ counters of the round-2 head_dim-64 attention kernels (dense cfg-2 shape, as profiles/r01_sq_counters.txt): separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r02sq; mkdir -p $O; rm -rf $O/p*
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/p$i -- python3 tools/pmc_flash.py > $O/p$i.log 2>&1
done
python tools/sq_summary.py $O/p1 $O/p2 $O/p3 $O/p4 | tee $O/r02_sq_counters_attention.txt
