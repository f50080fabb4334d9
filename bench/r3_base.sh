#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_base_bench1000.json 2> gpurun_out/r3_base_bench1000.err || exit 1
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --batch 1024 > gpurun_out/r3_base_bench1024.json 2> gpurun_out/r3_base_bench1024.err || exit 1
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --batch 512 > gpurun_out/r3_base_bench512.json 2> gpurun_out/r3_base_bench512.err || exit 1
timeout -k 10 300 python bench/conv_layers.py 512 fdw > gpurun_out/r3_base_conv_layers_b512.txt 2>&1 || exit 1
cat gpurun_out/r3_base_bench1000.json gpurun_out/r3_base_bench1024.json gpurun_out/r3_base_bench512.json
tail -3 gpurun_out/r3_base_conv_layers_b512.txt
