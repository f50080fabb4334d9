#!/usr/bin/env python3
"""Measured answer to "fold BN-apply(+ReLU) into the consumer conv's operand load" for the bn2 -> conv3 link of a
ResNet-50 bottleneck: [bn_apply(z) -> a ; conv1x1(a)] (two kernels, a written and re-read) against ONE conv kernel that
applies scale/shift/ReLU to the raw z fragments after the LDS read (csrc/conv_ring.hip, EPI bit 8).  Interleaved
timings in one process, random data, per-GPU batch 512."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_deep_learning_amd.ops import kernels as K

dev = torch.device("cuda:0")
CL = torch.channels_last


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    evs[0].record()
    for i in range(iters):
        fn()
        evs[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(iters))
    return ts[len(ts) // 2]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    print(f"{'link (bn2 -> conv3)':>28} | apply ms | conv ms | apply+conv | fused conv | fused/unfused | max err")
    for (C, HW, Kc) in [(64, 56, 256), (128, 28, 512), (256, 14, 1024), (512, 7, 2048)]:
        g = torch.Generator().manual_seed(1)
        z = (torch.randn((B, C, HW, HW), generator=g)).to(torch.bfloat16).to(dev).contiguous(memory_format=CL)
        w = (torch.randn((Kc, C, 1, 1), generator=g) / C ** 0.5).to(dev).contiguous(memory_format=CL)
        wk, _ = K.filter_repack(w)
        scale = (torch.rand(C, generator=g) + 0.5).to(dev)
        shift = (torch.randn(C, generator=g) * 0.3).to(dev)
        slab1, slab2 = K.new_slab(Kc, dev), K.new_slab(Kc, dev)

        def unfused():
            a, _m = K.bn_apply(z, scale, shift, relu=True, want_mask=True)
            return K.conv2d_fwd(a, wk, 1, 0, want_stats=True, slab=slab1)[0]

        def only_apply():
            return K.bn_apply(z, scale, shift, relu=True, want_mask=True)

        a0, _ = only_apply()

        def only_conv():
            return K.conv2d_fwd(a0, wk, 1, 0, want_stats=True, slab=slab1)[0]

        def fused():
            return K.conv1x1_fwd_inbn(z, wk, scale, shift, slab=slab2)[0]

        y0, y1 = unfused(), fused()
        err = (y0.float() - y1.float()).abs().max().item()
        s0, s1 = K.slab_sums(slab1), K.slab_sums(slab2)
        slab1.zero_(); slab2.zero_()
        ta, tc, tu, tf = timeit(only_apply), timeit(only_conv), timeit(unfused), timeit(fused)
        slab1.zero_(); slab2.zero_()
        print(f"  C{C:4d} {HW:2d}x{HW:<2d} -> K{Kc:4d} B{B:4d} | {ta:8.3f} | {tc:7.3f} | {tu:10.3f} | {tf:10.3f} | {tf / tu:13.3f} | "
              f"{err:.3g} (stats rel diff {((s0 - s1).abs().max() / s0.abs().max()).item():.2g})", flush=True)


if __name__ == "__main__":
    main()
