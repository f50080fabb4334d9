#!/usr/bin/env python3
"""Where does a filter-gradient kernel's time go?  Times `ddl_conv_wgrad` from variant builds of csrc/conv_wgrad.hip in
which parts of the halo kernel's main loop are compiled out (`-DHALO_ABL=bits`: 1 = no LDS-DMA requests inside the
loop, 2 = no MFMA, 4 = no transposing fragment reads), on ResNet-50's layer shapes.  The variants compute garbage; only
the timing is of interest.  Build (CPU box, before the GPU call; the variant id is the file name's number):

    cd distributed_deep_learning_amd/csrc && for n in 0 1 2 3 4 5 7; do hipcc -O3 -std=c++17 --offload-arch=gfx950 \
        -fPIC -fno-gpu-rdc -shared -DHALO_ABL=$n conv_wgrad.hip -o ../../bench/probes/bin/wg_abl$n.so; done

`DDL_WGRAD_HALO=0` in the environment times the ring kernels instead (every variant then runs the same code).

    python bench/wgrad_ablate.py [batch] [variant ids, comma separated: wg_abl<id>.so]
"""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SHAPES = [  # C, H, K, R, stride[, pad]
    (16, 115, 64, 4, 1, 0),          # the space-to-depth stem
    (64, 56, 64, 3, 1), (128, 28, 128, 3, 1), (256, 14, 256, 3, 1), (512, 7, 512, 3, 1),
    (256, 56, 64, 1, 1), (64, 56, 256, 1, 1), (1024, 14, 256, 1, 1), (256, 14, 1024, 1, 1), (512, 7, 2048, 1, 1),
]


def main() -> int:
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    ids = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(8))
    dev = torch.device("cuda:0")
    libs = {}
    for n in ids:
        path = os.path.join(HERE, "probes", "bin", f"wg_abl{n}.so")
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.ddl_conv_wgrad.restype = ctypes.c_int
            lib.ddl_conv_wgrad.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 11 + [ctypes.c_void_p, ctypes.c_longlong,
                                                                                         ctypes.c_void_p]
            libs[n] = lib
    names = {0: "full", 1: "-dma", 2: "-mfma", 3: "-dma-mfma", 4: "-reads", 5: "-dma-reads", 6: "-mfma-reads", 7: "none"}
    print(f"batch {batch}; ms per call, ring kernel variants", flush=True)
    print(f"{'shape':28s} " + " ".join(f"{names.get(n, 'v%d' % n):>10s}" for n in sorted(libs)), flush=True)
    st = torch.cuda.current_stream(dev).cuda_stream
    for shape in SHAPES:
        C, H, K, R, stride = shape[:5]
        pad = shape[5] if len(shape) > 5 else R // 2
        P = (H + 2 * pad - R) // stride + 1
        x = torch.randn(batch, H, H, C, device=dev).to(torch.bfloat16)
        dy = torch.randn(batch, P, P, K, device=dev).to(torch.bfloat16)
        dw = torch.zeros(K, R, R, C, device=dev)
        row = []
        for n in sorted(libs):
            lib = libs[n]

            def call():
                rc = lib.ddl_conv_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), batch, H, H, C, K, P, P, R, R, stride, pad,
                                        None, 0, st)
                assert rc == 0, rc
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / 10)
        print(f"C{C:5d} H{H:3d} K{K:5d} R{R} s{stride}    " + " ".join(f"{t:10.4f}" for t in row), flush=True)
        del x, dy, dw
    return 0


if __name__ == "__main__":
    sys.exit(main())
