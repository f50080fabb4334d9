#!/bin/bash
# Round-3 end evidence: per-launch listings of the final step at the benchmarked batch (serial and overlapped) and PMC
# counter groups per kernel family (batch 256: counter runs serialise the kernels).
set -u
OUT=gpurun_out; mkdir -p $OUT
bash bench/step_list.sh 1000 > /dev/null || exit 1
bash bench/pmc.sh r3final bench.py --steps 2 --warmup 2 --batch 256 || exit 1
for k in conv_ring_kernel conv_pring_kernel conv_big_kernel conv_wgrad_ring_kernel bn_bwd_apply_kernel bn_bwd_reduce_kernel bn_apply_kernel; do
  python bench/pmc_summary.py r3final $k
done > $OUT/pmc_r3final_summary.txt 2>&1
rm -rf $OUT/pmc_r3final_1 $OUT/pmc_r3final_2 $OUT/pmc_r3final_3
cat $OUT/pmc_r3final_summary.txt
