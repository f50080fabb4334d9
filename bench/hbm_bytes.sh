#!/bin/bash
# Fabric traffic of one headline step (serial: wgrad on the main stream, so every kernel has the chip to itself).
B=${1:-512}; OUT=gpurun_out; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
DDL_WGRAD_STREAM=0 timeout -s KILL 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $R/$OUT/hbm_pmc -- python3 $R/bench.py --steps 2 --warmup 2 --batch $B > $R/$OUT/hbm_pmc.log 2>&1 || { echo "pmc run failed"; tail -5 $R/$OUT/hbm_pmc.log; exit 1; }
cd $R
f=$(find $OUT/hbm_pmc -name '*counter_collection.csv' | head -1)
python bench/hbm_bytes.py "$f" > $OUT/hbm_bytes_b$B.txt 2>&1
rm -rf $OUT/hbm_pmc
cat $OUT/hbm_bytes_b$B.txt
