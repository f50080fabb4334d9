#!/bin/bash
set -u
for mode in 1 0 1 0; do
  export DDL_FUSED_FINALIZE=$mode
  timeout -k 10 300 python bench/ref_workloads.py --workloads cnn --batches 32,512,4096 --steps 200 --ours-only 2>&1 | grep "cnn" | sed "s/^/fused=$mode /" || exit 1
done
for mode in 1 0; do
  export DDL_FUSED_FINALIZE=$mode
  timeout -k 10 300 python bench.py --model resnet18_cifar --batch 256 --steps 50 --warmup 10 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | paste - - | sed "s/^/fused=$mode resnet18_cifar B=256 /"
done
