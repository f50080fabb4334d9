#!/bin/bash
# A/B of one environment switch over several values, alternating runs in one session.
#   bash bench/ab_env3.sh VAR BATCH v1 v2 v3 ...
VAR=$1; BATCH=$2; shift 2
OUT=gpurun_out; mkdir -p $OUT
for rep in 1 2; do
  for v in "$@"; do
    env $VAR=$v timeout -k 10 200 python bench.py --batch $BATCH --steps 30 --warmup 8 > $OUT/ab_env3.log 2>&1 || { echo "run failed ($VAR=$v)"; tail -5 $OUT/ab_env3.log; exit 1; }
    echo "$VAR=$v B=$BATCH $(grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' $OUT/ab_env3.log | tr '\n' ' ')"
  done
done
