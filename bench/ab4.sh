#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
for i in $(seq 1 12); do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --batch 512 > $OUT/ab4_$i.log 2>&1; rc=$?
  echo "$i exit $rc $(grep -o '"gpu_step_ms_min_median_max": [^]]*]' $OUT/ab4_$i.log) $(grep -o '"allocator_in_timed_region": {[^}]*}' $OUT/ab4_$i.log)"
  case $rc in 0) ;; *) exit $rc;; esac
done
for i in $(seq 1 10); do
  timeout -k 10 200 python _ab/old/bench.py --steps 20 --warmup 5 --batch 512 > $OUT/ab4_old_$i.log 2>&1; rc=$?
  echo "old $i exit $rc $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab4_old_$i.log)"
done
