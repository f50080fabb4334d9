#!/bin/bash
# Round-3 same-session evidence: stock PyTorch-ROCm ResNet-50 at the benchmarked batch, our bench, the ring-kernel footprint
# proxy table (what an RCCL ring kernel of N channels costs the compute streams), ResNet-18/CIFAR with forced collectives
# eager vs hipGraph.  Everything lands in gpurun_out/r3_evidence/.
set -o pipefail
O=gpurun_out/r3_evidence; mkdir -p $O
B=${1:-1000}
timeout -k 10 900 python -u bench/stock_resnet.py $B 20 > $O/stock_b$B.txt 2>&1 || { echo "stock failed"; tail -5 $O/stock_b$B.txt; exit 1; }
tail -1 $O/stock_b$B.txt
timeout -k 10 300 python -u bench.py --batch $B --steps 30 --warmup 10 > $O/bench_b$B.json 2>$O/bench_b$B.err || { echo "bench failed"; tail -5 $O/bench_b$B.err; exit 1; }
for wg in 0 16 32 64; do
  timeout -k 10 300 python -u bench.py --batch $B --steps 20 --warmup 8 --force-collectives --comm-proxy $wg > $O/proxy_$wg.json 2>$O/proxy_$wg.err || { echo "proxy $wg failed"; tail -5 $O/proxy_$wg.err; exit 1; }
  echo "proxy $wg done"
done
for g in off on; do
  timeout -k 10 200 python -u bench.py --model resnet18_cifar --batch 128 --steps 200 --warmup 20 --graph $g > $O/rn18_none_$g.json 2>$O/rn18_none_$g.err || { echo "rn18 none $g failed"; tail -5 $O/rn18_none_$g.err; exit 1; }
  timeout -k 10 200 python -u bench.py --model resnet18_cifar --batch 128 --steps 200 --warmup 20 --graph $g --force-collectives > $O/rn18_forced_$g.json 2>$O/rn18_forced_$g.err || { echo "rn18 forced $g failed"; tail -5 $O/rn18_forced_$g.err; exit 1; }
done
python - <<'PY'
import json, glob, os
O = "gpurun_out/r3_evidence"
for f in sorted(glob.glob(O + "/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    print(f"{os.path.basename(f):24s} {r['value']:10.1f} img/s {r['ms_per_step']:8.3f} ms/step graph={r['config']['hip_graph']} comm={r['config']['comm']} "
          f"proxy={r.get('comm_proxy_workgroups')} exposed={None if not r.get('comm_diagnostics') else r['comm_diagnostics']['exposed_comm_ms_per_step_median']}")
PY
