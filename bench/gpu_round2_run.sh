#!/bin/bash
# One GPU-box session of round 2: GPU test suite, multi-rank rehearsal over gloo, headline bench, launcher refusal,
# forced-collective overhead, then the (riskier) stream-priority probes.  Every GPU step has its own time limit; a step
# that times out, aborts or faults ends the session (no further GPU work is started).
set -u
OUT=gpurun_out
mkdir -p $OUT
step() {            # step NAME LIMIT cmd...   (exit codes 0/1/2/5 continue: test failures and expected refusals)
  local name=$1 limit=$2; shift 2
  echo "=== $name ===" | tee -a $OUT/r2_session.log
  timeout -k 10 "$limit" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "$name exit $rc" | tee -a $OUT/r2_session.log
  tail -n 6 "$OUT/$name.log"
  case $rc in 0|1|2|5) return 0;; *) echo "stopping the session after $name (exit $rc)" | tee -a $OUT/r2_session.log; exit $rc;; esac
}
: > $OUT/r2_session.log
for s in "$@"; do
  case $s in
    tests)   step r2_gputests 1100 python -u -m pytest tests -m gpu -q --maxfail 30 --timeout 400 --timeout-method thread ;;
    tests_new) step r2_gputests_new 900 python -u -m pytest tests/test_gpu_round2.py -m gpu -q --maxfail 30 --timeout 400 --timeout-method thread ;;
    gloo)    step r2_ddp_gloo_gpu_staged 300 python bench/ddp_gloo_gpu.py --transport staged ;;
    bench)   step r2_bench_b1024 300 python bench.py --steps 20 --warmup 5 ;;
    bench512) step r2_bench_b512 300 python bench.py --steps 20 --warmup 5 --batch 512 ;;
    refuse)  step r2_bench_gpus2_refused 120 python bench.py --gpus 2 --steps 2 --warmup 1 ;;
    forced)  step r2_bench_forced_native 300 python bench.py --steps 20 --warmup 5 --batch 512 --force-collectives
             step r2_bench_forced_torch 300 python bench.py --steps 20 --warmup 5 --batch 512 --force-collectives --comm torch ;;
    prio)    step r2_prio_normal 90 python bench/probes/prio_stream_probe.py normal
             step r2_prio_high 90 python bench/probes/prio_stream_probe.py high
             step r2_prio_high8 90 python bench/probes/prio_stream_probe.py high+8
             step r2_prio_highrnn 120 python bench/probes/prio_stream_probe.py high+rnn ;;
    gloodirect) step r2_ddp_gloo_gpu_direct 400 python bench/ddp_gloo_gpu.py --transport gloo-gpu --stall-s 40 ;;
    *) echo "unknown step $s"; exit 64;;
  esac
done
echo "session complete" | tee -a $OUT/r2_session.log
