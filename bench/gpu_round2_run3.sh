#!/bin/bash
# full GPU suite, smoke, the input-BN probe, then the profiled + un-profiled headline bench (B=512 and default)
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/r3_gputests.txt 2>&1 || { tail -30 $OUT/r3_gputests.txt; exit 1; }
tail -3 $OUT/r3_gputests.txt
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 10 200 python bench/inbn_probe.py 512 > $OUT/r3_inbn.txt 2>&1 || { tail $OUT/r3_inbn.txt; exit 1; }
cat $OUT/r3_inbn.txt
bash bench/prof_rn50.sh 512 r3_prof || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/r3_bench_default.log 2>&1 || { tail $OUT/r3_bench_default.log; exit 1; }
tail -1 $OUT/r3_bench_default.log
DDL_FUSED_FINALIZE=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --batch 512 > $OUT/r3_bench_nofuse.log 2>&1 || exit 1
echo "no fused finalize B=512: $(grep -o '"gpu_step_ms_min_median_max": [^]]*]' $OUT/r3_bench_nofuse.log)"
