#!/bin/bash
mkdir -p gpurun_out
show() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],2), round(j['value']), j['train_loss'], j['rows']['encoder_side'], j['rows']['decoder_side'], j['rows']['last_decoder_layer_query_side_and_heads'], round(j['step_mfma_frac'],4))"; }
timeout 600 python bench.py --no-cpu-baseline --no-probe --steps 20 --warmup 5 --force-reducer 2>gpurun_out/fr.err | show force-reducer || tail -5 gpurun_out/fr.err
timeout 600 python bench.py --no-cpu-baseline --no-probe --steps 20 --warmup 5 2>/dev/null | show plain
PB_PACK_ROWS=0 timeout 900 python bench.py --no-cpu-baseline --no-probe --steps 10 --warmup 3 --layers 24 --hs 1024 --ffn 4096 --heads 16 --seq 2048 --batch 8 2>gpurun_out/c5.err | show cfg5-padded || tail -5 gpurun_out/c5.err
timeout 900 python bench.py --no-cpu-baseline --no-probe --steps 10 --warmup 3 --layers 24 --hs 1024 --ffn 4096 --heads 16 --seq 2048 --batch 8 2>gpurun_out/c5p.err | show cfg5-packed || tail -5 gpurun_out/c5p.err
