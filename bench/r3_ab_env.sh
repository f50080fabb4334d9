#!/bin/bash
# usage: bench/r3_ab_env.sh VAR "values" "batches" [extra bench.py args]  -> headline step under VAR=value, alternating, one session
set -u
VAR=$1; VALS=$2; BATCHES=$3; shift 3
for B in $BATCHES; do
  for rep in 1 2; do
    for v in $VALS; do
      echo -n "B=$B $VAR=$v  "
      env $VAR=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --batch $B "$@" 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'], r['final_loss'])" || exit 1
    done
  done
done
