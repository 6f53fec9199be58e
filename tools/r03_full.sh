#!/bin/bash
# round 3: the whole GPU suite, smoke(), the driver-style bench line (stdout must be ONE JSON line)
mkdir -p gpurun_out/r03; O=gpurun_out/r03
timeout 2400 python -m pytest tests -x -q -m gpu > $O/full_tests.log 2>&1; tail -4 $O/full_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > $O/full_bench.json 2> $O/full_bench.err; wc -l $O/full_bench.json; python - <<'PY'
import json
lines=open('gpurun_out/r03/full_bench.json').read().strip().splitlines()
d=json.loads(lines[-1]); print(len(lines), 'line(s); ms/step', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],3), {k: (d[k].get('ms_per_step') or d[k].get('ms_per_token') or d[k]) for k in ('padded_step','dp_mode_step','decode')})
PY
