#!/usr/bin/env python3
"""Fused LSTM layer (csrc/lstm.hip) vs torch.nn.LSTM (MIOpen) on the reference workload's shape:
T=64 steps, 32 features, 128 hidden units (/root/reference/src/pytorch/LSTM/model.py:70-94).  Forward + backward."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_deep_learning_amd import nn as dnn


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    T, I, H = 64, 32, 128
    print(f"{'batch':>6} {'ours fwd+bwd ms':>16} {'MIOpen fp32 ms':>15} {'MIOpen bf16 ms':>15} {'speed-up vs fp32':>17}", flush=True)
    for B in (32, 256, 1024, 4096):
        ours = dnn.LSTM(I, H).cuda()
        stock = torch.nn.LSTM(I, H, batch_first=True).cuda()
        stock16 = torch.nn.LSTM(I, H, batch_first=True).cuda().to(torch.bfloat16)
        x = torch.randn(B, T, I, device="cuda", requires_grad=True)
        x16 = x.detach().to(torch.bfloat16).requires_grad_(True)

        def run(m, inp):
            out, _ = m(inp)
            out.float().sum().backward()

        a = timeit(lambda: run(ours, x))
        b = timeit(lambda: run(stock, x))
        try:
            c = timeit(lambda: run(stock16, x16))
        except RuntimeError:
            c = float("nan")
        print(f"{B:>6} {a:>16.3f} {b:>15.3f} {c:>15.3f} {b / a:>16.2f}x", flush=True)


if __name__ == "__main__":
    main()
