#!/bin/bash
# usage: bench/r3_big_dbg2.sh "dbg values" B  -> untimed-trace timing of conv_big (3x3 512 @7 forward) under each diagnostic switch
set -u
cd $GRAFT_REPO_ROOT
vals=$1; B=${2:-512}
for d in $vals; do
  echo -n "DDL_BIG_DBG=$d  "
  DDL_BIG_DBG=$d DDL_CONV_BIG=2 timeout -k 10 120 python - <<PY 2>&1 | grep -v amdgpu.ids || exit 1
import os, sys, torch
sys.path.insert(0, os.getcwd())
from distributed_deep_learning_amd.ops import kernels as K
dev = torch.device("cuda:0")
out = []
for (C, H, Kc, R, st, pad) in [(512, 7, 512, 3, 1, 1), (256, 14, 256, 3, 1, 1), (256, 14, 1024, 1, 1, 0)]:
    x = torch.randn($B, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Kc, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    wk, wt = K.filter_repack(w)
    slab = K.new_slab(Kc, dev)
    f = lambda: K.conv2d_fwd(x, wk, st, pad, want_stats=True, slab=slab)
    f(); f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    out.append(s.elapsed_time(e) / 10)
print("  ".join(f"{t:.4f}" for t in out))
PY
done
