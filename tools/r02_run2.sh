# round 2, GPU call: full -m gpu suite, bench (default), decode bench + kernel trace, cfg-5-shape bench, fc1 PMC passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -rfE -p no:cacheprovider > $O/t2.log 2>&1; echo "pytest rc=$?" >> $O/t2.log; tail -4 $O/t2.log
timeout 600 python bench.py > $O/bench2.json 2> $O/bench2.err; python -c "import json;r=json.load(open('$O/bench2.json'));print(r['ms_per_step'],r['value'],r['roofline']['frac'],r['roofline']['families_in_step'])"
timeout 600 python bench.py --mode decode --steps 512 --no-cpu-baseline > $O/decode2.json 2> $O/decode2.err; cat $O/decode2.json
rm -rf $O/prof_dec; timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof_dec -- python3 bench.py --mode decode --no-cpu-baseline --steps 200 > $O/prof_dec.log 2>&1
python tools/rocpd_decode.py $(ls $O/prof_dec/*/*.db | head -1) > $O/decode_stats.txt; cat $O/decode_stats.txt
timeout 600 python bench.py --layers 24 --hs 1024 --ffn 4096 --heads 16 --seq 2048 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-probe > $O/bench_cfg5.json 2> $O/bench_cfg5.err; python -c "import json;r=json.load(open('$O/bench_cfg5.json'));print('cfg5',r['ms_per_step'],r['value'],r['step_mfma_frac'])"
rm -rf $O/pmc_f $O/pmc_w
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_f -- python3 tools/pmc_gemm.py > $O/pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_w -- python3 tools/pmc_gemm.py > $O/pmc_w.log 2>&1
ls $O/pmc_f/*/ $O/pmc_w/*/ 2>/dev/null | head
