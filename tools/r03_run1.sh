#!/bin/bash
# round 3, run 1: new tests first (fast feedback), the whole GPU suite, default bench with the extra legs
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "graph_decode or kv_cache or cfg2_size or g8" > $O/run1_decode_tests.log 2>&1; tail -5 $O/run1_decode_tests.log
timeout 600 python -m pytest tests/test_packed_gpu.py tests/test_pretrain_gpu.py -x -q -m gpu -k "oracle_directly or demo_midi" -s > $O/run1_new_tests.log 2>&1; tail -15 $O/run1_new_tests.log
PB_PROBE_DUMP=$O/run1_probe.txt python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/run1_bench.json 2> $O/run1_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/run1_bench.json').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'frac', d['step_mfma_frac'])
for k in ('padded_step','dp_mode_step','decode'): print(k, json.dumps(d.get(k))[:400])
PY
tail -3 $O/run1_bench.err
timeout 1800 python -m pytest tests -x -q -m gpu > $O/run1_tests.log 2>&1; tail -4 $O/run1_tests.log
