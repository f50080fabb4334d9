#!/bin/bash
# A/B of one environment switch inside the headline step, alternating runs in one session.
#   bash bench/ab_env.sh VAR VALUE_A VALUE_B [BATCH] [extra bench.py args...]
VAR=$1; A=$2; Bv=$3; BATCH=${4:-512}; shift 4 2>/dev/null || shift $#
OUT=gpurun_out; mkdir -p $OUT
for rep in 1 2; do
  for v in $A $Bv; do
    env $VAR=$v timeout -k 10 200 python bench.py --batch $BATCH --steps 30 --warmup 8 "$@" > $OUT/ab_env_$v.log 2>&1 || { echo "run failed ($VAR=$v)"; tail -5 $OUT/ab_env_$v.log; exit 1; }
    echo "$VAR=$v B=$BATCH $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"gpu_step_ms_min_median_max": [^]]*]\|"final_loss": [0-9.]*' $OUT/ab_env_$v.log | tr '\n' ' ')"
  done
done
