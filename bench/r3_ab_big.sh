#!/bin/bash
# usage: bench/r3_ab_big.sh "batches"  -> headline step with conv_big off / on, alternating, one session
set -u
cd $GRAFT_REPO_ROOT
for B in $1; do
  for rep in 1 2; do
    for v in ${2:-0 1}; do
      echo -n "B=$B DDL_CONV_BIG=$v  "
      DDL_CONV_BIG=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --batch $B 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'], r['final_loss'])" || exit 1
    done
  done
done
