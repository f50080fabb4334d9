#!/bin/bash
# usage: bench/r3_pmc_big.sh C H K R stride pad mode [B]  -> counters of one conv layer under the ring kernels and under conv_big
set -u
cd $GRAFT_REPO_ROOT
export DDL_ONE_B=${8:-512}
for v in 0 2; do
  export DDL_CONV_BIG=$v
  bash bench/pmc.sh big$v bench/conv_one.py "${@:1:7}" || exit 1
  python bench/pmc_summary.py big$v conv_ > gpurun_out/pmc_big${v}_summary.txt 2>&1
  cat gpurun_out/pmc_big${v}_summary.txt
  rm -rf gpurun_out/pmc_big${v}_[123]
done
