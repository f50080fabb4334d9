#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 300 python -u -m pytest tests/test_gpu_round2.py -m gpu -q -x -k "side_stream or linear or l1" --timeout 200 --timeout-method thread 2>&1 | tail -3
for b in 1024 1536 2048 1024 2048; do
  timeout -k 10 300 python bench.py --steps 15 --warmup 4 --batch $b > $OUT/sweep_$b.log 2>&1; rc=$?
  echo "B=$b exit $rc $(grep -o '"value": [0-9.]*' $OUT/sweep_$b.log) $(grep -o '"gpu_step_ms_min_median_max": [^]]*]' $OUT/sweep_$b.log) $(grep -o '"reserved_gib": [0-9.]*' $OUT/sweep_$b.log)"
  case $rc in 0) ;; *) tail -5 $OUT/sweep_$b.log; exit $rc;; esac
done
