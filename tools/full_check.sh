#!/bin/bash
# What the driver runs at round end, in one gpurun call: the whole GPU suite, smoke(), the default bench line (stdout must be ONE JSON line).
#   gpurun --timeout 3000 -- 'bash tools/full_check.sh r03'
TAG=${1:-check}; O=gpurun_out/$TAG; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/full_tests.log 2>&1; tail -4 $O/full_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > $O/full_bench.json 2> $O/full_bench.err; O=$O python - <<'PY'
import json, os
lines = open(os.environ['O'] + '/full_bench.json').read().strip().splitlines()
d = json.loads(lines[-1])
print(len(lines), 'line(s); ms/step', round(d['ms_per_step'], 2), 'frac', round(d['roofline']['frac'], 3),
      {k: (d[k].get('ms_per_step') or d[k].get('ms_per_token') or d[k]) for k in ('padded_step', 'real_mix_step', 'dp_mode_step', 'decode')})
PY
