#!/bin/bash
# usage: bench/r3_pmc_insts.sh DBG C H K R stride pad mode [B]  -> instruction-mix counters of conv_big for one layer
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export DDL_BIG_DBG=$1; shift
export DDL_ONE_B=${8:-512} DDL_CONV_BIG=2
rm -rf gpurun_out/pmc_insts_1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_insts_1 -- python3 bench/conv_one.py "${@:1:7}" > gpurun_out/pmc_insts.log 2>&1 || { echo failed; tail -5 gpurun_out/pmc_insts.log; exit 1; }
python bench/pmc_summary.py insts conv_big
rm -rf gpurun_out/pmc_insts_1
