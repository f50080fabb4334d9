"""The space-to-depth stem conv (4 x 4 taps x 16 channels -> 64, 115 x 115 -> 112 x 112) in isolation: forward + statistics
and filter gradient, time per launch and parity against fp32.

    [DDL_CONV_STEM=0|1] [DDL_WGRAD_HALO=0|1] python bench/stem_probe.py [B]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from distributed_deep_learning_amd.ops import kernels as K  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


g = torch.Generator().manual_seed(0)
for (C, H, Kc, R, st, pad) in [(16, 115, 64, 4, 1, 0), (16, 57, 64, 4, 1, 1), (8, 40, 64, 4, 1, 2), (32, 30, 128, 2, 1, 0)]:
    n = B if H > 100 else 16
    x = torch.randn((n, C, H, H), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Kc, C, R, R), generator=g) / (C * R * R) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    wk, _ = K.filter_repack(w)
    slab = K.new_slab(Kc, dev)
    y, _ = K.conv2d_fwd(x, wk, st, pad, want_stats=True, slab=slab)
    ref = F.conv2d(x[:4].float(), wk.permute(0, 3, 1, 2).float(), None, st, pad)
    err = float((y[:4].float() - ref).abs().max() / ref.abs().max())
    s = K.slab_sums(slab).double()
    yf = y.float().double()
    serr = float(((s[0] - yf.sum((0, 2, 3))).abs() / (yf.abs().sum((0, 2, 3)) + 1e-9)).max())
    tf = timeit(lambda: K.conv2d_fwd(x, wk, st, pad, want_stats=True, slab=slab))
    dy = torch.randn_like(y)
    tw = timeit(lambda: K.conv2d_wgrad(dy, x, R, R, st, pad))
    print(f"C{C:3d} H{H:4d} K{Kc:4d} R{R} pad{pad} n{n:4d}: fwd+stats {tf:.4f} ms  wgrad {tw:.4f} ms   fwd err {err:.4f}  sum err {serr:.2e}",
          flush=True)
