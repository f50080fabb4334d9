"""Pieces of the LSTM layer's backward outside the recurrent kernel: input gradient dG @ W_ih on rows_gemm_nt against torch.mm,
weight gradients on the split-M GEMM; per batch size of the reference workload (T = 64, I = 32, H = 128).

    python bench/lstm_bwd_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from distributed_deep_learning_amd.ops import kernels as K  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print("us per call:   rows_gemm_nt | torch.mm |  gemm_tn dW_ih  | gemm_tn dW_hh")
for B in (32, 512, 4096):
    M, H4, I, H = B * 64, 512, 32, 128
    g2 = torch.randn(M, H4, device=dev).to(torch.bfloat16)
    w = torch.randn(H4, I, device=dev).to(torch.bfloat16)
    wt = w.t().contiguous()
    x2 = torch.randn(M, I, device=dev).to(torch.bfloat16)
    h2 = torch.randn(M, H, device=dev).to(torch.bfloat16)
    dwi = torch.zeros(H4, I, device=dev)
    dwh = torch.zeros(H4, H, device=dev)
    a = timeit(lambda: K.rows_gemm_nt(g2, wt))
    b = timeit(lambda: torch.mm(g2, w))
    c = timeit(lambda: K.gemm_tn(g2, x2, dwi))
    d = timeit(lambda: K.gemm_tn(g2, h2, dwh))
    err = float((K.rows_gemm_nt(g2, wt).float() - torch.mm(g2, w).float()).abs().max())
    print(f"B={B:5d} M={M:7d}:  {a:8.1f} | {b:8.1f} | {c:8.1f} | {d:8.1f}   (max diff {err:.3f})", flush=True)
