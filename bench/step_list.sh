#!/bin/bash
# Per-launch listing of one headline step: serial (wgrad on the main stream) and overlapped.  B from $1 (default 512).
set -u
B=${1:-512}; shift || true      # further arguments go to bench.py (e.g. --model vgg16)
OUT=gpurun_out; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
DDL_WGRAD_STREAM=0 timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/sl_serial -- python3 $R/bench.py --steps 3 --warmup 2 --batch $B "$@" > $R/$OUT/sl_serial.log 2>&1 || { echo "serial trace failed"; exit 1; }
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/sl_over -- python3 $R/bench.py --steps 3 --warmup 2 --batch $B "$@" > $R/$OUT/sl_over.log 2>&1 || { echo "overlapped trace failed"; exit 1; }
cd $R
python bench/trace_gaps.py "$(find $OUT/sl_serial -name '*kernel_trace.csv' | head -1)" 40 list > $OUT/step_list_serial_b$B.txt
python bench/trace_gaps.py "$(find $OUT/sl_over -name '*kernel_trace.csv' | head -1)" 40 list > $OUT/step_list_overlapped_b$B.txt
rm -rf $OUT/sl_serial $OUT/sl_over
head -45 $OUT/step_list_serial_b$B.txt
