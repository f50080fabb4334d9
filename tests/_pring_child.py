"""Child process of test_gpu_round2.py::test_persistent_ring_kernel_*: the kernel-selection switches are read once per
process, so the persistent ring kernel (csrc/conv_pring.hip) is forced here (DDL_CONV_PRING / DDL_PRING_MIN_TILES from the
parent's environment) and compared with plain fp32 PyTorch on shapes with >= 512 tiles and ragged M / OC tails."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_deep_learning_amd.ops import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
N = 67
# C, H, K, R, stride, pad, kind: fa = forward + statistics, d = dgrad, da = dgrad + masked addend, dn = dgrad + plain addend
CASES = [(64, 45, 192, 1, 1, 0, "fa"), (96, 33, 320, 3, 1, 1, "fa"), (128, 47, 512, 3, 2, 1, "fa"), (64, 45, 256, 1, 1, 0, "fa"),
         (192, 45, 64, 1, 1, 0, "da"), (320, 33, 96, 1, 1, 0, "da"), (256, 45, 64, 1, 1, 0, "dn"), (320, 33, 96, 3, 1, 1, "d")]
failed = 0
g = torch.Generator().manual_seed(5)
for (C, H, Kc, R, st, pad, kind) in CASES:
    x = (torch.randn(N, C, H, H, generator=g) * 1.5 + 0.5).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Kc, C, R, R, generator=g) / (C * R * R) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    wk, wt = K.filter_repack(w)
    wq = wk.permute(0, 3, 1, 2).float()
    ok = True
    if kind == "fa":
        for rep in range(2):                                   # the slab is left for the caller to fold: twice = accumulates twice
            slab = K.new_slab(Kc, dev)
            y, _ = K.conv2d_fwd(x, wk, st, pad, want_stats=True, slab=slab)
            ref = F.conv2d(x.float(), wq, None, st, pad)
            ok &= bool(((y.float() - ref).abs().max() <= 2e-2 * ref.abs().max()).item())
            s = K.slab_sums(slab).double()
            yf = y.float().double()
            r1, r2 = yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))
            ok &= bool(((s[0] - r1).abs() <= 1e-4 * r1.abs() + 1e-2).all().item())
            ok &= bool(((s[1] - r2).abs() <= 1e-5 * r2.abs() + 1e-2).all().item())
    else:
        y = F.conv2d(x.float(), wq, None, st, pad)
        dy = torch.randn(y.shape, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ref = torch.nn.grad.conv2d_input(x.shape, wq, dy.float(), st, pad)
        if kind in ("da", "dn"):
            add = torch.randn(x.shape, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            if kind == "da":
                mask = torch.randint(0, 256, (N * H * H, C // 8), generator=g, dtype=torch.uint8).to(dev)
                bits = ((mask.unsqueeze(-1) >> torch.arange(8, device=dev)) & 1).reshape(N, H, H, C).permute(0, 3, 1, 2)
                ref = ref + add.float() * bits.float()
                dx = K.conv2d_dgrad(dy, wt, (H, H), st, pad, addend=add, addend_mask=mask)
            else:
                ref = ref + add.float()
                dx = K.conv2d_dgrad(dy, wt, (H, H), st, pad, addend=add)
        else:
            dx = K.conv2d_dgrad(dy, wt, (H, H), st, pad)
        ok = bool(((dx.float() - ref).abs().max() <= 2e-2 * ref.abs().max()).item())
    print(f"{'ok  ' if ok else 'FAIL'} {kind} C={C} H={H} K={Kc} R={R} s={st}", flush=True)
    failed += 0 if ok else 1
torch.cuda.synchronize()
print(f"{failed} failed")
sys.exit(1 if failed else 0)
