#!/bin/bash
# usage: bench/pmc.sh <tag> <python args...>   -> gpurun_out/pmc_<tag>_*.csv   (one counter group per run)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1 || { echo "pmc group $i failed"; tail -3 gpurun_out/pmc_${tag}_$i.log; exit 1; }
done
echo done
