#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
run() { tag=$1; shift; for i in 1 2 3 4 5 6 7 8; do
  timeout -k 10 200 "$@" > $OUT/ab3_${tag}_$i.log 2>&1; rc=$?
  echo "$tag $i exit $rc $(grep -o '"gpu_step_ms_min_median_max": [^]]*]' $OUT/ab3_${tag}_$i.log)"
  case $rc in 0) ;; *) exit $rc;; esac
done; }
A="--steps 20 --warmup 5 --batch 512"
DDL_LINEAR_ATEN=1 run aten python bench.py $A
DDL_WGRAD_STREAM=0 run noside python bench.py $A
run default python bench.py $A
