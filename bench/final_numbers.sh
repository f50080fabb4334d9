#!/bin/bash
# One session producing the numbers quoted in README.md / docs/STATUS.md (ours and the stock comparators on the SAME box).
set -u
OUT=gpurun_out; mkdir -p $OUT
F=$OUT/final_numbers.txt; : > $F
line() { grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"reserved_gib": [0-9.]*' | paste -sd' ' ; }
run() { local tag=$1; shift; local o; o=$(timeout -k 10 400 python bench.py --steps 20 --warmup 5 "$@" 2>&1 | tail -1 | line); rc=$?; echo "$tag | $o" | tee -a $F; }
run "resnet50 B=1000 (default)"
run "resnet50 B=1024" --batch 1024
run "resnet50 B=512" --batch 512
run "resnet50 B=256" --batch 256
run "resnet50 B=512 forced collectives native" --batch 512 --force-collectives --comm native
run "resnet50 B=512 forced collectives c10d" --batch 512 --force-collectives --comm torch
run "resnet50 B=512 LARS + ps sync forced collectives" --batch 512 --force-collectives --optimizer lars --sync ps
run "resnet50 B=512 bf16 wire forced collectives" --batch 512 --force-collectives --grad-dtype bf16
run "resnet18_cifar B=512" --model resnet18_cifar --batch 512
run "vgg16 B=128" --model vgg16 --batch 128
for m in "resnet50 512" "resnet18_cifar 512" "vgg16 128"; do
  set -- $m
  timeout -k 10 900 python bench/stock_models.py $1 $2 > $OUT/final_stock_$1.log 2>&1 &   # heartbeat lines keep the log growing
  wait $!
  echo "stock $1 B=$2 | $(tail -1 $OUT/final_stock_$1.log)" | tee -a $F
done
timeout -k 10 900 python bench/ref_workloads.py --steps 100 > $OUT/final_ref_workloads.log 2>&1
grep -v amdgpu.ids $OUT/final_ref_workloads.log | tee -a $F
