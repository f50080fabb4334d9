#!/bin/bash
# Round-end evidence: per-launch listings of the final step (serial and overlapped) and PMC counter groups per kernel family.
set -u
OUT=gpurun_out; mkdir -p $OUT
bash bench/step_list.sh 512 > /dev/null || exit 1
bash bench/pmc.sh final2 bench.py --steps 2 --warmup 2 --batch 256 || exit 1
for k in conv_ring_kernel conv_pring_kernel conv_wgrad_ring_kernel bn_bwd_apply_kernel bn_bwd_reduce_kernel bn_apply_kernel; do
  python bench/pmc_summary.py final2 $k
done > $OUT/pmc_final2_summary.txt 2>&1
rm -rf $OUT/pmc_final2_1 $OUT/pmc_final2_2 $OUT/pmc_final2_3
cat $OUT/pmc_final2_summary.txt
