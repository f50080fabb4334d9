#!/bin/bash
# A/B of the fused BatchNorm finalize (last-arriver fold inside the producing kernel) in ONE session: traces + bench
set -u
OUT=gpurun_out; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
for mode in 1 0 1 0; do
  export DDL_FUSED_FINALIZE=$mode
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --batch 512 > $OUT/fin_ab_bench_$mode.log 2>&1 || exit 1
  echo "fused=$mode $(grep -o '"gpu_step_ms_min_median_max": [^]]*]' $OUT/fin_ab_bench_$mode.log)"
done
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  export DDL_FUSED_FINALIZE=$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/fin_ab_$mode -- python3 $R/bench.py --steps 4 --warmup 2 --batch 512 > $R/$OUT/fin_ab_$mode.log 2>&1 || exit 1
done
