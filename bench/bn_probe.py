"""Effective bandwidth of the BN kernels vs a plain device copy, over ResNet-50's activation shapes.

    python bench/bn_probe.py [batch]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from distributed_deep_learning_amd.ops import kernels as K  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(64, 112), (64, 56), (256, 56), (128, 28), (512, 28), (256, 14), (1024, 14), (512, 7), (2048, 7)]
CL = torch.channels_last


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


print(f"{'shape':>16} {'MB':>7} | copy TB/s | apply  TB/s | apply+res TB/s | bwd ms  TB/s | bwd+dres TB/s")
for C, H in SHAPES:
    x = torch.randn(B, C, H, H, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=CL)
    r = torch.randn_like(x, memory_format=CL)
    dy = torch.randn_like(x, memory_format=CL)
    y = torch.empty_like(x, memory_format=CL)
    nb = x.numel() * 2
    gamma = torch.ones(C, device="cuda")
    beta = torch.zeros(C, device="cuda")
    slab = K.bn_stats(x)
    vec = K.bn_finalize(slab, B * H * H, gamma, beta, None, None, 0.1, 1e-5)
    _, mask = K.bn_apply(x, vec[2], vec[3], relu=True, want_mask=True)
    t_copy = timeit(lambda: y.copy_(x))
    t_app = timeit(lambda: K.bn_apply(x, vec[2], vec[3], relu=True, want_mask=True))
    t_res = timeit(lambda: K.bn_apply(x, vec[2], vec[3], residual=r, relu=True, want_mask=True))
    t_bwd = timeit(lambda: K.bn_backward(dy, mask, x, gamma, vec[0], vec[1], relu=True, slab=slab))
    t_bwd2 = timeit(lambda: K.bn_backward(dy, mask, x, gamma, vec[0], vec[1], relu=True, want_dres=True, slab=slab))
    mb = nb / 1e6
    tb = lambda nbytes, t: nbytes / t / 1e12
    print(f"C{C:5d} H{H:4d}     {mb:7.1f} | {tb(2 * nb, t_copy):8.2f}  | {t_app * 1e3:6.3f} {tb(2.0625 * nb, t_app):5.2f} |"
          f" {t_res * 1e3:6.3f} {tb(3.0625 * nb, t_res):5.2f}   | {t_bwd * 1e3:6.3f} {tb(5.125 * nb, t_bwd):5.2f} |"
          f" {t_bwd2 * 1e3:6.3f} {tb(6.125 * nb, t_bwd2):5.2f}", flush=True)
