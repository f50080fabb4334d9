#!/bin/bash
# Why is the forced-collective step 2x slower through the native communicator than through torch.distributed?
set -u
OUT=gpurun_out; mkdir -p $OUT
run() { name=$1; shift; echo "=== $name"; timeout -k 10 200 "$@" > $OUT/$name.log 2>&1; rc=$?; echo "$name exit $rc"; grep -o '"ms_per_step": [0-9.]*' $OUT/$name.log; case $rc in 0) ;; *) exit $rc;; esac; }
B="--steps 10 --warmup 4 --batch 256 --force-collectives"
run nsp_torch python bench.py $B --comm torch
run nsp_native python bench.py $B --comm native
run nsp_native_nowd python bench.py $B --comm native --watchdog-s 0
DDL_WGRAD_STREAM=0 run nsp_native_noside python bench.py $B --comm native
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/nsp_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --batch 256 --force-collectives --comm native > $GRAFT_REPO_ROOT/$OUT/nsp_prof.log 2>&1
echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT
f=$(find $OUT/nsp_prof -name '*kernel_stats.csv' | head -1); echo $f; head -15 "$f"
