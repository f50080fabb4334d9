#!/bin/bash
# same-session A/B of this tree against a build of an earlier commit under _ab/prev (interleaved, B=512)
set -u
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
  for t in . _ab/prev; do
    cd $R/$t
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --batch 512 2>&1 | grep -o '"gpu_step_ms_min_median_max": [^]]*]' | sed "s|^|$t |" || exit 1
  done
done
