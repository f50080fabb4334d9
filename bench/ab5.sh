#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
A=${1:-"--steps 20 --warmup 5 --batch 512"}; N=${2:-12}
for i in $(seq 1 $N); do
  timeout -k 10 200 python bench.py $A > $OUT/ab5_$i.log 2>&1; rc=$?
  echo "$i exit $rc $(grep -o '"gpu_step_ms_min_median_max": [^]]*]' $OUT/ab5_$i.log) $(grep -o '"allocator_in_timed_region": {[^}]*}' $OUT/ab5_$i.log)"
  case $rc in 0) ;; *) tail -5 $OUT/ab5_$i.log; exit $rc;; esac
done
