#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 200 "$@" > $OUT/$name.log 2>&1; rc=$?; echo "$name exit $rc $(grep -o '"ms_per_step": [0-9.]*' $OUT/$name.log)"; case $rc in 0) ;; *) exit $rc;; esac; }
B="--steps 20 --warmup 5 --batch 512 --force-collectives"
run nsp2_native_a python bench.py $B --comm native
run nsp2_torch_a python bench.py $B --comm torch
run nsp2_native_b python bench.py $B --comm native
run nsp2_torch_b python bench.py $B --comm torch
run nsp2_native_1024 python bench.py --steps 20 --warmup 5 --batch 1024 --force-collectives --comm native
run nsp2_plain_512 python bench.py --steps 20 --warmup 5 --batch 512
