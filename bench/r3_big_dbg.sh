#!/bin/bash
# usage: bench/r3_big_dbg.sh "dbg values" B C H K R stride pad mode  -> phase trace of conv_big under each diagnostic switch
set -u
cd $GRAFT_REPO_ROOT
vals=$1; shift
for d in $vals; do
  echo "=== DDL_BIG_DBG=$d"
  DDL_BIG_DBG=$d timeout -k 10 120 python bench/big_trace.py "$@" 2>&1 | grep -v amdgpu.ids || exit 1
done
