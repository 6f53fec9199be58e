#!/bin/bash
# round 3 baseline on one box: GPU tests + default bench with the full in-step op table
O=gpurun_out/r03; mkdir -p $O
PB_PROBE_DUMP=$O/base_probe.txt python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/base_bench.json 2> $O/base_bench.err
tail -c 600 $O/base_bench.json; echo
head -50 $O/base_probe.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/base_tests.log 2>&1; tail -3 $O/base_tests.log
