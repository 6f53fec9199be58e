# full regression: build check, smoke, -m gpu suite, default bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r02; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q -rfE -p no:cacheprovider > $O/t3.log 2>&1; echo "pytest rc=$?" >> $O/t3.log; tail -4 $O/t3.log
timeout 600 python bench.py > $O/bench3.json 2> $O/bench3.err; python -c "import json;r=json.load(open('$O/bench3.json'));print(r['ms_per_step'],r['value'],r['roofline']['frac'],r['cpu_baseline'])"
