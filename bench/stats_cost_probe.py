"""What do the BatchNorm statistics in the conv epilogue cost?  Forward conv with and without `want_stats` on the 1x1 layers
with wide outputs (the layers whose forward runs far below its data gradient's rate), ring kernels only.

    python bench/stats_cost_probe.py [B]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from distributed_deep_learning_amd.ops import kernels as K  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


print(f"B={B}: ms per launch   no stats | stats | copy of the output tensor (write+read at stream rate)")
for (C, H, Kc, R, st, pad) in [(64, 56, 256, 1, 1, 0), (128, 28, 512, 1, 1, 0), (256, 14, 1024, 1, 1, 0), (512, 7, 2048, 1, 1, 0),
                               (256, 56, 64, 1, 1, 0), (1024, 14, 256, 1, 1, 0), (256, 14, 256, 3, 1, 1)]:
    x = torch.randn(B, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Kc, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    wk, _ = K.filter_repack(w)
    slab = K.new_slab(Kc, dev)
    y, _ = K.conv2d_fwd(x, wk, st, pad)
    t0 = timeit(lambda: K.conv2d_fwd(x, wk, st, pad))
    t1 = timeit(lambda: K.conv2d_fwd(x, wk, st, pad, want_stats=True, slab=slab))
    y2 = torch.empty_like(y)
    t2 = timeit(lambda: y2.copy_(y))
    print(f"C{C:5d} H{H:3d} K{Kc:5d} R{R}   {t0:.4f} | {t1:.4f} | {t2:.4f}   out {y.numel() * 2 / 1e6:.0f} MB", flush=True)
