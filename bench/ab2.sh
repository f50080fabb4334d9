#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --batch 512 > $OUT/ab2_new_$i.log 2>&1; rc=$?
  echo "new $i exit $rc $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab2_new_$i.log) $(grep -o '"gpu_step_ms_min_median_max": [^]]*]' $OUT/ab2_new_$i.log)"
  case $rc in 0) ;; *) exit $rc;; esac
done
