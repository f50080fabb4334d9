"""Per-phase shader-cycle breakdown of the big-tile ping-pong conv kernel (csrc/conv_big.hip, TRACE build).

    DDL_CONV_BIG=2 python bench/big_trace.py B C H K R stride pad [f|d]

Every wave sums s_memtime deltas per phase: load-phase issue (12 fragment reads + the stage's LDS-DMA), counted vmcnt
wait, lgkmcnt + barrier behind the load phase, MFMA phase issue, barrier behind the MFMA phase, epilogue."""
import ctypes
import os
import sys

os.environ.setdefault("DDL_CONV_BIG", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from distributed_deep_learning_amd.ops import _lib  # noqa: E402
from distributed_deep_learning_amd.ops import kernels as K  # noqa: E402

B, C, H, Kc, R, st, pad = [int(v) for v in sys.argv[1:8]]
mode = sys.argv[8] if len(sys.argv) > 8 else "f"
dev = torch.device("cuda:0")
x = torch.randn(B, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(Kc, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
wk, wt = K.filter_repack(w)
y, _ = K.conv2d_fwd(x, wk, st, pad, want_stats=True)
dy = torch.randn_like(y)
lib = _lib.require()
lib.ddl_conv_set_trace.argtypes = [ctypes.c_void_p]
lib.ddl_conv_set_trace.restype = None
buf = torch.zeros(256 * 8, 16, dtype=torch.int64, device=dev)


def run():
    if mode == "f":
        K.conv2d_fwd(x, wk, st, pad, want_stats=True)
    else:
        K.conv2d_dgrad(dy, wt, (H, H), st, pad)


for _ in range(3):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    run()
e.record()
torch.cuda.synchronize()
print(f"untraced: {s.elapsed_time(e) / 5:.4f} ms per launch")
lib.ddl_conv_set_trace(ctypes.c_void_p(buf.data_ptr()))
s.record()
run()
e.record()
torch.cuda.synchronize()
lib.ddl_conv_set_trace(None)
print(f"traced:   {s.elapsed_time(e):.4f} ms")
t = buf.cpu().double()
t = t[t[:, 15] > 0]
steps = t[:, 15]
names = {0: "sub-step 0: fragment reads", 2: "sub-step 0: lgkm + barrier", 3: "sub-step 0: MFMA phase (+DMA issue)",
         4: "sub-step 0: barrier", 6: "sub-step 1: fragment reads", 7: "sub-step 1: vmcnt wait", 8: "sub-step 1: lgkm + barrier",
         9: "sub-step 1: MFMA phase (+DMA, +vmcnt)", 10: "sub-step 1: barrier", 5: "epilogue (per tile sum)"}
print(f"waves {len(t)}  K64-stages per wave: mean {steps.mean():.1f}  total cycles per wave: mean {t[:, 14].mean():.0f}")
for g in (0, 1):
    sel = t[(torch.arange(len(t)) % 8 // 4) == g]
    print(f" group {g}:")
    for i in (0, 2, 3, 4, 6, 7, 8, 9, 10, 5):
        per = sel[:, i] / sel[:, 15]
        print(f"   {names[i]:40s} {per.mean():8.1f} cycles per stage  (p10 {per.quantile(0.1):7.1f}  p90 {per.quantile(0.9):7.1f})")
    print(f"   {'sum / stage':40s} {(sel[:, 14] / sel[:, 15]).mean():8.1f}")
