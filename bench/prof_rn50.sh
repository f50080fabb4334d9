#!/bin/bash
# rocprofv3 kernel trace of the headline step (B from $1, default 512) + per-step summary; then an un-profiled bench run.
set -u
B=${1:-512}; TAG=${2:-prof}
OUT=gpurun_out; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/$TAG -- python3 $R/bench.py --steps 4 --warmup 2 --batch $B > $R/$OUT/$TAG.log 2>&1; echo "rocprof exit $?"
cd $R
f=$(find $OUT/$TAG -name '*kernel_trace.csv' | head -1)
python bench/trace_gaps.py "$f" 30 | tee $OUT/${TAG}_summary.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --batch $B > $OUT/${TAG}_bench.log 2>&1; echo "bench exit $? $(grep -o '"gpu_step_ms_min_median_max": [^]]*]' $OUT/${TAG}_bench.log)"
