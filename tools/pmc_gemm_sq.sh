cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r05g; mkdir -p $O; rm -rf $O/p*
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/p$i -- python3 tools/pmc_gemm.py --M=26624 --plain > $O/p$i.log 2>&1
done
python - <<'PY'
import csv, glob
tot={}
for f in glob.glob('gpurun_out/r05g/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm3' not in r['Kernel_Name']: continue
        e=tot.setdefault(r['Counter_Name'],[0.0,0]); e[0]+=float(r['Counter_Value']); e[1]+=1
for k,v in sorted(tot.items()): print('%-28s %14.4g  (%d dispatches)'%(k, v[0]/v[1], v[1]))
PY
