#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r02g; mkdir -p $O; rm -rf $O/prof
rocprofv3 --kernel-trace -d $R/$O/prof -- python3 bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 2 > $O/prof.log 2>&1
DB=$(ls $O/prof/*/*.db | head -1)
python tools/rocpd_overlap.py $DB --gaps --tail
rm -rf $O/prof/*/*.db
