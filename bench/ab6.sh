#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
A=${1:-"--steps 20 --warmup 5 --batch 512"}; N=${2:-4}
for i in $(seq 1 $N); do
  for pol in ${POLS:-hold join record_stream}; do
    DDL_SIDE_LIFETIME=$pol timeout -k 10 200 python bench.py $A > $OUT/ab6_${pol}_$i.log 2>&1; rc=$?
    echo "$pol $i exit $rc $(grep -o '"gpu_step_ms_min_median_max": [^]]*]' $OUT/ab6_${pol}_$i.log) $(grep -o '"reserved_gib": [0-9.]*' $OUT/ab6_${pol}_$i.log)"
    case $rc in 0) ;; *) exit $rc;; esac
  done
done
