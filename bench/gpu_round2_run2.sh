#!/bin/bash
# GPU session: new tests, then model benches (ResNet-50, VGG-16) and their kernel summaries.
set -u
OUT=gpurun_out; mkdir -p $OUT
step() { local name=$1 limit=$2; shift 2; echo "=== $name"; timeout -k 10 "$limit" "$@" > "$OUT/$name.log" 2>&1; local rc=$?; echo "$name exit $rc"; tail -n 4 "$OUT/$name.log"; case $rc in 0|1|2|5) return 0;; *) echo "stopping after $name"; exit $rc;; esac; }
step r2b_tests_new 900 python -u -m pytest tests/test_gpu_round2.py tests/test_gpu_kernels.py -m gpu -q --maxfail 30 --timeout 400 --timeout-method thread
step r2b_tests_model 900 python -u -m pytest tests/test_gpu_model.py tests/test_s2d.py -m gpu -q --maxfail 30 --timeout 400 --timeout-method thread
step r2b_bench_resnet50_b512 300 python bench.py --steps 20 --warmup 5 --batch 512
step r2b_bench_vgg16_b128 300 python bench.py --steps 20 --warmup 5 --batch 128 --model vgg16
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/r2b_prof_vgg -- python3 $R/bench.py --steps 4 --warmup 2 --batch 128 --model vgg16 > $R/$OUT/r2b_prof_vgg.log 2>&1; echo "prof vgg exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/r2b_prof_rn50 -- python3 $R/bench.py --steps 4 --warmup 2 --batch 512 > $R/$OUT/r2b_prof_rn50.log 2>&1; echo "prof rn50 exit $?"
cd $R
for d in r2b_prof_vgg r2b_prof_rn50; do f=$(find $OUT/$d -name '*kernel_stats.csv' | head -1); echo "== $f"; head -40 "$f" | cut -c1-160; done
