#!/bin/bash
# A/B of the persistent ring kernel inside the headline step: same session, alternating runs.  B from $1 (default 512).
B=${1:-512}; OUT=gpurun_out; mkdir -p $OUT
for rep in 1 2; do
  for v in 0 1; do
    DDL_CONV_PRING=$v timeout -k 10 200 python bench.py --batch $B --steps 30 --warmup 8 > $OUT/ab_pring_$v.log 2>&1 || { echo "run failed ($v)"; tail -5 $OUT/ab_pring_$v.log; exit 1; }
    echo "DDL_CONV_PRING=$v B=$B $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"gpu_step_ms_min_median_max": [^]]*]\|"final_loss": [0-9.]*' $OUT/ab_pring_$v.log | tr '\n' ' ')"
  done
done
