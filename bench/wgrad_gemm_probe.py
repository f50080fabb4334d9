"""Is a library GEMM (hipBLASLt through torch.mm) faster than conv_wgrad for the plain 1x1 layers?  dw[K,C] = dy^T x."""
import sys
import time

import torch

sys.path.insert(0, ".")
from distributed_deep_learning_amd.ops import kernels as K  # noqa: E402

B = 256
dev = torch.device("cuda:0")
SHAPES = [(64, 56, 64), (64, 56, 256), (256, 56, 64), (256, 56, 128), (128, 28, 512), (512, 28, 128), (512, 28, 256),
          (256, 14, 1024), (1024, 14, 256), (1024, 14, 512), (512, 7, 2048), (2048, 7, 512)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for C, H, Kc in SHAPES:
    M = B * H * H
    x = torch.randn(B, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Kc, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x2 = x.permute(0, 2, 3, 1).reshape(M, C)
    dy2 = dy.permute(0, 2, 3, 1).reshape(M, Kc)
    out = torch.zeros(Kc, C, 1, 1, device=dev).contiguous(memory_format=torch.channels_last)
    t_ours = timeit(lambda: K.conv2d_wgrad(dy, x, 1, 1, 1, 0, out=out))
    t_mm = timeit(lambda: torch.mm(dy2.t(), x2))
    try:
        t_mm32 = timeit(lambda: torch.mm(dy2.t(), x2, out_dtype=torch.float32))
    except Exception as e:  # noqa: BLE001
        t_mm32 = float("nan")
        err = str(e)[:60]
    gf = 2.0 * M * C * Kc / 1e9
    print(f"C{C:5d} H{H:3d} K{Kc:5d}  {gf:6.1f} GF | ours {t_ours:.3f} ms {gf / t_ours:6.0f} TF/s | mm bf16 {t_mm:.3f} ms "
          f"{gf / t_mm:6.0f} | mm->fp32 {t_mm32:.3f} ms {gf / t_mm32:6.0f}", flush=True)
