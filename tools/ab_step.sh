#!/bin/bash
# Same-box A/B of the whole pre-train step between two builds of the library:
#   cp pianobart_amd/libpianobart_hip.so ab/old.so   (before the change; ab/ travels with the gpurun snapshot, keep it out of git)
#   ... edit, python pianobart_amd/build.py ...
#   gpurun -- 'bash tools/ab_step.sh'
for r in 1 2; do
  for v in old new; do
    if [ "$v" = old ]; then export PB_LIB_PATH=$PWD/ab/old.so; else unset PB_LIB_PATH; fi
    python bench.py --no-cpu-baseline --steps 10 --warmup 3 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', round(r['ms_per_step'],2), 'ms/step')"
  done
done
