#!/bin/bash
# The pre-train step at the other shapes of BASELINE.json / SURVEY 8: configs[4]'s model on one GPU (24L/1024d/ffn4096/h16, S = 2048,
# B = 8) and the reference's CLI-default shape (12L/768d/ffn2048/8 heads = head_dim 96), next to configs[1]. One line each.
#   gpurun -- 'bash tools/other_shapes.sh'
run() { python bench.py --no-cpu-baseline --no-probe --steps 10 --warmup 4 "$@" 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$*', '| ms/step', round(r['ms_per_step'], 2), '| tokens/s', round(r['value']), '| step_mfma_frac', round(r['step_mfma_frac'], 3), '| rows', r['rows']['encoder_side'], r['rows']['decoder_side'])"; }
run
run --layers 24 --hs 1024 --ffn 4096 --heads 16 --seq 2048 --batch 8
run --ffn 2048 --heads 8
