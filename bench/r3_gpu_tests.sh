#!/bin/bash
# the whole GPU test tier in one process, output under gpurun_out/
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 ${2:-900} python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread ${1:+-k "$1"} > gpurun_out/gpu_tests.txt 2>&1
rc=$?
tail -${3:-25} gpurun_out/gpu_tests.txt
exit $rc
