#!/bin/bash
# A/B of the single-buffer dense block against layer-by-layer concatenation on the reference CNN (kernel summaries + times)
set -u
OUT=gpurun_out; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
for mode in 1 0; do
  export DDL_DENSE_BUFFER=$mode
  timeout -k 10 200 python bench/ref_workloads.py --workloads cnn --batches 32,512,4096 --steps 100 --ours-only 2>&1 | grep cnn | sed "s/^/buffer=$mode /" || exit 1
done
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  export DDL_DENSE_BUFFER=$mode
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/dense_ab_$mode -- python3 $R/bench/ref_workloads.py --workloads cnn --batches 4096 --steps 20 --ours-only > $R/$OUT/dense_ab_$mode.log 2>&1 || exit 1
  f=$(find $R/$OUT/dense_ab_$mode -name '*kernel_stats.csv' | head -1)
  echo "== buffer=$mode kernel stats (top 28)"; head -29 "$f" | cut -d, -f1-4,7 
done
