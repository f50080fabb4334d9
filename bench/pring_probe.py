"""Persistent ring conv kernel (csrc/conv_pring.hip) against the ring kernel: parity with an fp32 reference and time per
launch over the layer classes it takes (forward + statistics, plain dgrad, dgrad + masked addend).

    python bench/pring_probe.py [B]        # parent: one child with DDL_CONV_PRING=0, one with 1
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [
    ("fwd+stats  64->256 @56", (64, 56, 256, 1, 1, 0), "fa"),
    ("fwd+stats 128->512 @28", (128, 28, 512, 1, 1, 0), "fa"),
    ("fwd+stats 256->1024 @14", (256, 14, 1024, 1, 1, 0), "fa"),
    ("fwd+stats 512->2048 @7", (512, 7, 2048, 1, 1, 0), "fa"),
    ("fwd+stats 1024->256 @14", (1024, 14, 256, 1, 1, 0), "fa"),
    ("fwd+stats 256->128 @56", (256, 56, 128, 1, 1, 0), "fa"),
    ("fwd+stats 3x3 128 @28", (128, 28, 128, 3, 1, 1), "fa"),
    ("fwd+stats 3x3 256 @14", (256, 14, 256, 3, 1, 1), "fa"),
    ("fwd+stats 3x3 s2 128 @56", (128, 56, 128, 3, 2, 1), "fa"),
    ("dgrad+addend 256<-64 @56", (256, 56, 64, 1, 1, 0), "da"),
    ("dgrad+addend 512<-128 @28", (512, 28, 128, 1, 1, 0), "da"),
    ("dgrad+addend 1024<-256 @14", (1024, 14, 256, 1, 1, 0), "da"),
    ("dgrad+addend 2048<-512 @7", (2048, 7, 512, 1, 1, 0), "da"),
    ("dgrad 256<-1024 @14", (256, 14, 1024, 1, 1, 0), "d"),
    ("dgrad 128<-512 @28", (128, 28, 512, 1, 1, 0), "d"),
    ("dgrad 3x3 256 @14", (256, 14, 256, 3, 1, 1), "d"),
    ("dgrad 3x3 128 @28", (128, 28, 128, 3, 1, 1), "d"),
]


def child(B):
    import torch
    import torch.nn.functional as F
    from distributed_deep_learning_amd.ops import kernels as K
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

    # parity at a small batch with ragged tails (M % 256 != 0, OC % 128 != 0), forced through the kernel under test
    errs = []
    for (C, H, Kc, R, st, pad, kind) in [(64, 45, 192, 1, 1, 0, "fa"), (96, 33, 320, 3, 1, 1, "fa"), (192, 45, 64, 1, 1, 0, "da"),
                                         (320, 33, 96, 3, 1, 1, "d"), (128, 47, 512, 3, 2, 1, "fa"), (320, 33, 96, 1, 1, 0, "da")]:
        Bp = 67          # >= 512 tiles with ragged M and OC tails: the persistent kernel takes every case
        x = (torch.randn(Bp, C, H, H, device=dev) + 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Kc, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        wk, wt = K.filter_repack(w)
        wb = w.to(torch.bfloat16).float()
        if kind == "fa":
            slab = K.new_slab(Kc, dev)
            y, _ = K.conv2d_fwd(x, wk, st, pad, want_stats=True, slab=slab)
            ref = F.conv2d(x.float(), wb, stride=st, padding=pad)
            e1 = ((y.float() - ref).abs().max() / ref.abs().max()).item()
            sums = K.slab_sums(slab).double()
            yf = y.float().double()
            rs = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))])
            e2 = ((sums[:2].reshape(2, -1)[:, :Kc] - rs).abs() / rs.abs().clamp_min(1.0)).max().item()
            errs.append((kind, C, H, Kc, R, st, round(e1, 5), round(e2, 7)))
        else:
            y = F.conv2d(x.float(), wb, stride=st, padding=pad)
            dy = torch.randn_like(y).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            ref = torch.nn.grad.conv2d_input(x.shape, wb, dy.float(), stride=st, padding=pad)
            if kind == "da":
                add = torch.randn_like(x)
                mask = torch.randint(0, 256, (Bp * H * H, C // 8), dtype=torch.uint8, device=dev)
                bits = ((mask.unsqueeze(-1) >> torch.arange(8, device=dev)) & 1).reshape(Bp, H, H, C).permute(0, 3, 1, 2)
                ref = ref + add.float() * bits.float()
                dx = K.conv2d_dgrad(dy, wt, (H, H), st, pad, addend=add, addend_mask=mask)
            else:
                dx = K.conv2d_dgrad(dy, wt, (H, H), st, pad)
            e1 = ((dx.float() - ref).abs().max() / ref.abs().max()).item()
            errs.append((kind, C, H, Kc, R, st, round(e1, 5)))
    print("PARITY " + repr(errs), flush=True)

    out = []
    for name, (C, H, Kc, R, st, pad), kind in CASES:
        x = torch.randn(B, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Kc, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        wk, wt = K.filter_repack(w)
        y, _ = K.conv2d_fwd(x, wk, st, pad)
        if kind == "fa":
            slab = K.new_slab(Kc, dev)
            t = timeit(lambda: K.conv2d_fwd(x, wk, st, pad, want_stats=True, slab=slab))
        else:
            dy = torch.randn_like(y)
            if kind == "da":
                add = torch.randn_like(x)
                mask = torch.randint(0, 256, (B * H * H, C // 8), dtype=torch.uint8, device=dev)
                t = timeit(lambda: K.conv2d_dgrad(dy, wt, (H, H), st, pad, addend=add, addend_mask=mask))
            else:
                t = timeit(lambda: K.conv2d_dgrad(dy, wt, (H, H), st, pad))
        out.append(t)
    print("RESULT " + " ".join(f"{t:.4f}" for t in out), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "child":
        child(int(sys.argv[1]))
        return
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    rows = []
    settings = [("ring", {"DDL_CONV_PRING": "0"}), ("pring", {"DDL_CONV_PRING": "2"}),
                ("ring", {"DDL_CONV_PRING": "0"}), ("pring", {"DDL_CONV_PRING": "2"})]
    for name, e in settings:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(B), "child"], env=dict(os.environ, **e),
                           capture_output=True, text=True, timeout=280)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
        par = [ln for ln in r.stdout.splitlines() if ln.startswith("PARITY")]
        print(name, par[0] if par else "(no parity line)", flush=True)
        if not line:
            print(f"{name} failed (rc {r.returncode}):\n{r.stdout[-1500:]}\n{r.stderr[-2500:]}")
            sys.exit(1)
        rows.append([float(v) for v in line[0].split()[1:]])
    print(f"B={B}; ms per launch: " + "  ".join(n for n, _ in settings) + "  | pring/ring (second pair)")
    for i, (name, _, _) in enumerate(CASES):
        print(f"{name:30s} " + " ".join(f"{rows[j][i]:7.4f}" for j in range(len(settings))) + f"  | {rows[3][i] / rows[2][i]:5.2f}")
    print(f"{'sum':30s} " + " ".join(f"{sum(rows[j]):7.4f}" for j in range(len(settings))) + f"  | {sum(rows[3]) / sum(rows[2]):5.2f}")


if __name__ == "__main__":
    main()
