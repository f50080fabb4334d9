#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for t in cur prev; do
  d=$R; [ $t = prev ] && d=$R/_ab/prev
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/pab_$t -- python3 $d/bench.py --steps 4 --warmup 2 --batch 512 > $OUT/pab_$t.log 2>&1 || exit 1
done
