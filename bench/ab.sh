#!/bin/bash
# Interleaved A/B of two trees in one GPU session: bench/ab.sh "<args for bench.py>" [rounds]
set -u
OUT=gpurun_out; mkdir -p $OUT
ARGS=${1:-"--steps 20 --warmup 5 --batch 512"}; N=${2:-3}
for i in $(seq 1 $N); do
  for t in old new; do
    if [ $t = old ]; then B=_ab/old/bench.py; else B=bench.py; fi
    timeout -k 10 200 python $B $ARGS > $OUT/ab_${t}_$i.log 2>&1; rc=$?
    echo "$t $i exit $rc $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_${t}_$i.log)"
    case $rc in 0) ;; *) tail -5 $OUT/ab_${t}_$i.log; exit $rc;; esac
  done
done
