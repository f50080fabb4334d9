import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_deep_learning_amd.ops import kernels as K
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B = 256
for (C, H, Kc, R, st, pad) in [(64, 56, 256, 1, 1, 0), (256, 56, 64, 1, 1, 0), (256, 14, 1024, 1, 1, 0), (1024, 14, 256, 1, 1, 0), (256, 14, 256, 3, 1, 1), (64, 56, 64, 3, 1, 1)]:
    x = torch.randn(B, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Kc, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    wk, wt = K.filter_repack(w)
    slab = K.new_slab(Kc, dev)
    y, _ = K.conv2d_fwd(x, wk, st, pad)
    t_ns = timeit(lambda: K.conv2d_fwd(x, wk, st, pad))
    t_st = timeit(lambda: K.conv2d_fwd(x, wk, st, pad, want_stats=True, slab=slab))
    y2 = torch.empty_like(y)
    t_cp = timeit(lambda: y2.copy_(y))
    t_rd = timeit(lambda: K.bn_stats(x, slab=slab))
    mb = (x.numel() + y.numel()) * 2 / 1e6
    print(f"C{C} H{H} K{Kc} R{R}: no-stats {t_ns:6.1f} us  stats {t_st:6.1f} us | copy(y) {t_cp:6.1f} us  read(x) {t_rd:6.1f} us | io {mb:.0f} MB -> {mb/t_ns:.2f} TB/s", flush=True)
