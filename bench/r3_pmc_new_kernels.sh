#!/bin/bash
# counters of the kernels added at the end of round 3 (halo filter gradient, stem forward) inside a batch-256 step
set -u
OUT=gpurun_out; mkdir -p $OUT
bash bench/pmc.sh r3new bench.py --steps 2 --warmup 2 --batch 512 || exit 1
for k in conv_wgrad_halo_kernel conv_stem_fwd_kernel conv_wgrad_ring_kernel; do
  python bench/pmc_summary.py r3new $k
done > $OUT/pmc_r3new_summary.txt 2>&1
rm -rf $OUT/pmc_r3new_1 $OUT/pmc_r3new_2 $OUT/pmc_r3new_3
cat $OUT/pmc_r3new_summary.txt
