#!/bin/bash
# Full-step A/B of the wide (128 x 256) filter-gradient tiles: "WIDE PAD CAP" triples, alternating runs in one session.
OUT=gpurun_out; mkdir -p $OUT
B=${1:-512}
for rep in 1 2; do
  for t in "0 0 256" "1 0 256" "1 1 256" "1 0 320" "1 1 224"; do
    set -- $t
    DDL_WGRAD_WIDE=$1 DDL_WGRAD_WIDE_PAD=$2 DDL_WGRAD_WIDE_CAP=$3 timeout -k 10 200 python bench.py --batch $B --steps 30 --warmup 8 > $OUT/ab_caps.log 2>&1 || { echo "run failed"; tail -5 $OUT/ab_caps.log; exit 1; }
    echo "wide=$1 pad=$2 cap=$3 $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_caps.log)"
  done
done
