"""Per-layer conv micro-benchmark over the ResNet-50 shape list: time and TFLOP/s of fwd / dgrad / wgrad, and how far
each is from its ROOFLINE  max(FLOP / 2.5 PF/s dense bf16 peak, compulsory bytes / 5.3 TB/s)  — 5.3 TB/s being what a
plain copy kernel reaches on this part (profiles/r1_hbm_stream_probe.txt).  Compulsory bytes: every operand and the
result cross HBM exactly once (bf16 activations / filters, fp32 filter gradient).  A layer at 0.15 PF/s that sits at
85 % of its roofline is bandwidth-bound, not slow."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_deep_learning_amd.ops import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
which = sys.argv[2] if len(sys.argv) > 2 else "fdw"
# (C, H, K, R, stride, pad, count)
SHAPES = [
    (8, 224, 64, 7, 2, 3, 1),
    (64, 56, 64, 1, 1, 0, 1), (64, 56, 64, 3, 1, 1, 3), (64, 56, 256, 1, 1, 0, 4), (256, 56, 64, 1, 1, 0, 2),
    (256, 56, 128, 1, 1, 0, 1), (128, 56, 128, 3, 2, 1, 1), (128, 28, 512, 1, 1, 0, 4), (256, 56, 512, 1, 2, 0, 1),
    (512, 28, 128, 1, 1, 0, 3), (128, 28, 128, 3, 1, 1, 3),
    (512, 28, 256, 1, 1, 0, 1), (256, 28, 256, 3, 2, 1, 1), (256, 14, 1024, 1, 1, 0, 6), (512, 28, 1024, 1, 2, 0, 1),
    (1024, 14, 256, 1, 1, 0, 5), (256, 14, 256, 3, 1, 1, 5),
    (1024, 14, 512, 1, 1, 0, 1), (512, 14, 512, 3, 2, 1, 1), (512, 7, 2048, 1, 1, 0, 3), (1024, 14, 2048, 1, 2, 0, 1),
    (2048, 7, 512, 1, 1, 0, 2), (512, 7, 512, 3, 1, 1, 2),
]
dev = torch.device("cuda:0")

def timeit(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

PEAK, BW = 2.5e15, 5.3e12


def roof_ms(flop, nbytes):
    return max(flop / PEAK, nbytes / BW) * 1e3


tot = {"f": 0.0, "d": 0.0, "w": 0.0}
roof = {"f": 0.0, "d": 0.0, "w": 0.0}
print(f"{'shape':38s} {'GFLOP':>8s} | fwd ms  TF/s %roof | dgrad ms TF/s %roof | wgrad ms TF/s %roof | bound")
for (C, H, Kc, R, st, pad, cnt) in SHAPES:
    x = torch.randn(B, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Kc, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    wk, wt = K.filter_repack(w)
    y, _ = K.conv2d_fwd(x, wk, st, pad)
    dy = torch.randn_like(y)
    P = y.shape[2]
    gf = 2.0 * B * P * P * Kc * C * R * R / 1e9
    row = f"C{C:4d} H{H:3d} K{Kc:4d} R{R} s{st} x{cnt}".ljust(38) + f" {gf:8.1f} |"
    act = 2.0 * (x.numel() + y.numel())
    wbytes = 2.0 * w.numel()
    r_act = roof_ms(gf * 1e9, act + wbytes)                 # fwd and dgrad move the same tensors
    r_w = roof_ms(gf * 1e9, act + 2 * wbytes)               # fp32 filter gradient
    if "f" in which:
        t = timeit(lambda: K.conv2d_fwd(x, wk, st, pad, want_stats=True)); tot["f"] += t * cnt; roof["f"] += r_act * cnt
        row += f" {t:6.3f} {gf / t:5.0f} {100 * r_act / t:4.0f}% |"
    if "d" in which and C != 8:
        t = timeit(lambda: K.conv2d_dgrad(dy, wt, (H, H), st, pad)); tot["d"] += t * cnt; roof["d"] += r_act * cnt
        row += f" {t:6.3f} {gf / t:5.0f}  {100 * r_act / t:4.0f}% |"
    elif "d" in which:
        row += "    -      -      - |"
    if "w" in which:
        t = timeit(lambda: K.conv2d_wgrad(dy, x, R, R, st, pad)); tot["w"] += t * cnt; roof["w"] += r_w * cnt
        row += f" {t:6.3f} {gf / t:5.0f}  {100 * r_w / t:4.0f}% |"
    row += " HBM" if (act + wbytes) / BW > gf * 1e9 / PEAK else " MFMA"
    print(row, flush=True)
print("total ms per step:", {k: round(v, 2) for k, v in tot.items()})
print("roofline ms per step:", {k: round(v, 2) for k, v in roof.items()},
      "-> achieved fraction", {k: round(roof[k] / tot[k], 3) for k in tot if tot[k] > 0})
