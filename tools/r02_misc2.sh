#!/bin/bash
# other shapes, padded vs packed step (same box): head_dim 96 (12L/768d/ffn3072, 8 heads) and the reference's CLI default (8L/1024d/ffn2048, 8 heads = head_dim 128)
show() { grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(j['ms_per_step'],2), 'tok/s', round(j['value']), 'loss', round(j['train_loss'],5), 'rows', j['rows']['encoder_side'], j['rows']['decoder_side'], j['rows']['last_decoder_layer_query_side_and_heads'])"; }
for pk in 0 1; do
  PB_PACK_ROWS=$pk timeout 600 python bench.py --no-cpu-baseline --no-probe --steps 20 --warmup 5 --heads 8 2>/dev/null | show "hd96 pack=$pk"
  PB_PACK_ROWS=$pk timeout 600 python bench.py --no-cpu-baseline --no-probe --steps 20 --warmup 5 --layers 8 --hs 1024 --ffn 2048 --heads 8 2>/dev/null | show "cli-default pack=$pk"
done
