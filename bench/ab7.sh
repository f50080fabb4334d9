#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
A="--steps 15 --warmup 4 --batch 1024"
run() { tag=$1; shift; timeout -k 10 300 env "$@" python bench.py $A > $OUT/ab7_$tag.log 2>&1; rc=$?; echo "$tag exit $rc $(grep -o '"gpu_step_ms_min_median_max": [^]]*]' $OUT/ab7_$tag.log) $(grep -o '"reserved_gib": [0-9.]*' $OUT/ab7_$tag.log)"; case $rc in 0) ;; *) tail -3 $OUT/ab7_$tag.log;; esac; }
for i in 1 2; do
run hold_$i DDL_SIDE_LIFETIME=hold
run hold_split32_$i DDL_SIDE_LIFETIME=hold PYTORCH_HIP_ALLOC_CONF=max_split_size_mb:32 PYTORCH_CUDA_ALLOC_CONF=max_split_size_mb:32
run keep2_$i DDL_SIDE_LIFETIME=hold DDL_SIDE_KEEP_JOINS=2
run keep2_split32_$i DDL_SIDE_LIFETIME=hold DDL_SIDE_KEEP_JOINS=2 PYTORCH_HIP_ALLOC_CONF=max_split_size_mb:32 PYTORCH_CUDA_ALLOC_CONF=max_split_size_mb:32
done
