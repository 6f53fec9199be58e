#!/bin/bash
# Same-box A/B of the whole pre-train step between library builds (PB_LIB_PATH), alternating, two rounds:  bash tools/ab_libs.sh ab/head.so ab/new.so ...
# (build them with PB_LIB_OUT=... [PB_CSRC=...] python pianobart_amd/build.py; boxes differ by +-3 %: never compare across gpurun calls)
for r in 1 2; do
  for lib in "$@"; do
    PB_LIB_PATH=$PWD/$lib python bench.py --no-cpu-baseline --no-probe --steps 20 --warmup 8 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$lib', round(r['ms_per_step'],2), round(r['ms_per_step_median_hip_events'],2), r['train_loss'])"
  done
done
