"""Big-tile ping-pong conv kernel (csrc/conv_big.hip) against the ring / persistent-ring kernels: parity with an fp32
reference on ragged shapes, then time per launch over ResNet-50's layer classes.

    python bench/big_probe.py [B] [parity|time|both]   # parent: children with DDL_CONV_BIG=0 and 2
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [
    ("fwd+stats 3x3  64->64  @56", (64, 56, 64, 3, 1, 1), "fa"),
    ("fwd+stats 3x3 128 @28", (128, 28, 128, 3, 1, 1), "fa"),
    ("fwd+stats 3x3 256 @14", (256, 14, 256, 3, 1, 1), "fa"),
    ("fwd+stats 3x3 512 @7", (512, 7, 512, 3, 1, 1), "fa"),
    ("fwd+stats 3x3 s2 128 @56", (128, 56, 128, 3, 2, 1), "fa"),
    ("fwd+stats 3x3 s2 256 @28", (256, 28, 256, 3, 2, 1), "fa"),
    ("fwd+stats 3x3 s2 512 @14", (512, 14, 512, 3, 2, 1), "fa"),
    ("fwd+stats  64->256 @56", (64, 56, 256, 1, 1, 0), "fa"),
    ("fwd+stats 128->512 @28", (128, 28, 512, 1, 1, 0), "fa"),
    ("fwd+stats 256->1024 @14", (256, 14, 1024, 1, 1, 0), "fa"),
    ("fwd+stats 512->2048 @7", (512, 7, 2048, 1, 1, 0), "fa"),
    ("fwd+stats 256->128 @56", (256, 56, 128, 1, 1, 0), "fa"),
    ("fwd+stats 512->128 @28", (512, 28, 128, 1, 1, 0), "fa"),
    ("fwd+stats 512->256 @28", (512, 28, 256, 1, 1, 0), "fa"),
    ("fwd+stats 1024->256 @14", (1024, 14, 256, 1, 1, 0), "fa"),
    ("fwd+stats 1024->512 @14", (1024, 14, 512, 1, 1, 0), "fa"),
    ("fwd+stats 2048->512 @7", (2048, 7, 512, 1, 1, 0), "fa"),
    ("dgrad 3x3 128 @28", (128, 28, 128, 3, 1, 1), "d"),
    ("dgrad 3x3 256 @14", (256, 14, 256, 3, 1, 1), "d"),
    ("dgrad 3x3 512 @7", (512, 7, 512, 3, 1, 1), "d"),
    ("dgrad 128<-512 @28", (128, 28, 512, 1, 1, 0), "d"),
    ("dgrad 256<-1024 @14", (256, 14, 1024, 1, 1, 0), "d"),
    ("dgrad 512<-2048 @7", (512, 7, 2048, 1, 1, 0), "d"),
    ("dgrad+addend 256<-64 @56", (256, 56, 64, 1, 1, 0), "da"),
    ("dgrad+addend 512<-128 @28", (512, 28, 128, 1, 1, 0), "da"),
    ("dgrad+addend 1024<-256 @14", (1024, 14, 256, 1, 1, 0), "da"),
    ("dgrad+addend 2048<-512 @7", (2048, 7, 512, 1, 1, 0), "da"),
]

PARITY = [(64, 45, 192, 1, 1, 0, "fa"), (96, 33, 320, 3, 1, 1, "fa"), (128, 47, 512, 3, 2, 1, "fa"), (64, 45, 256, 1, 1, 0, "fa"),
          (256, 21, 256, 3, 1, 1, "fa"), (32, 19, 128, 1, 1, 0, "fa"), (128, 33, 320, 3, 1, 1, "fa"), (192, 25, 1024, 1, 1, 0, "fa"),
          (320, 33, 128, 1, 1, 0, "da"), (320, 33, 128, 3, 1, 1, "d"), (1024, 25, 192, 1, 1, 0, "d"),
          # 530 tiles = 2 rounds + 18: the launcher cuts the batch (64 images big-tile, 3 images ring kernels)
          (128, 45, 256, 3, 1, 1, "fa"), (256, 45, 128, 3, 1, 1, "d"), (256, 45, 128, 1, 1, 0, "da"),
          (192, 45, 64, 1, 1, 0, "da"), (320, 33, 96, 1, 1, 0, "da"), (256, 45, 64, 1, 1, 0, "dn"), (320, 33, 96, 3, 1, 1, "d"),
          (256, 21, 256, 3, 1, 1, "d")]


def parity(Bp=67):
    import torch
    import torch.nn.functional as F
    from distributed_deep_learning_amd.ops import kernels as K
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    failed = 0
    for (C, H, Kc, R, st, pad, kind) in PARITY:
        x = (torch.randn(Bp, C, H, H, generator=g) * 1.5 + 0.5).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Kc, C, R, R, generator=g) / (C * R * R) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
        wk, wt = K.filter_repack(w)
        wq = wk.permute(0, 3, 1, 2).float()
        ok = True
        detail = ""
        if kind == "fa":
            for rep in range(2):
                slab = K.new_slab(Kc, dev)
                y, _ = K.conv2d_fwd(x, wk, st, pad, want_stats=True, slab=slab)
                ref = F.conv2d(x.float(), wq, None, st, pad)
                e = ((y.float() - ref).abs().max() / ref.abs().max()).item()
                ok &= e <= 2e-2
                s = K.slab_sums(slab).double()
                yf = y.float().double()
                r1, r2 = yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))
                ok &= bool(((s[0] - r1).abs() <= 1e-4 * r1.abs() + 1e-2).all().item())
                ok &= bool(((s[1] - r2).abs() <= 1e-5 * r2.abs() + 1e-2).all().item())
                detail = f"err {e:.4f} sums {((s[0] - r1).abs().max()).item():.3g} {((s[1] - r2).abs() / r2.abs()).max().item():.3g}"
        else:
            y = F.conv2d(x.float(), wq, None, st, pad)
            dy = torch.randn(y.shape, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            ref = torch.nn.grad.conv2d_input(x.shape, wq, dy.float(), st, pad)
            if kind in ("da", "dn"):
                add = torch.randn(x.shape, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                if kind == "da":
                    mask = torch.randint(0, 256, (Bp * H * H, C // 8), generator=g, dtype=torch.uint8).to(dev)
                    bits = ((mask.unsqueeze(-1) >> torch.arange(8, device=dev)) & 1).reshape(Bp, H, H, C).permute(0, 3, 1, 2)
                    ref = ref + add.float() * bits.float()
                    dx = K.conv2d_dgrad(dy, wt, (H, H), st, pad, addend=add, addend_mask=mask)
                else:
                    ref = ref + add.float()
                    dx = K.conv2d_dgrad(dy, wt, (H, H), st, pad, addend=add)
            else:
                dx = K.conv2d_dgrad(dy, wt, (H, H), st, pad)
            e = ((dx.float() - ref).abs().max() / ref.abs().max()).item()
            ok = e <= 2e-2
            detail = f"err {e:.4f}"
        print(f"{'ok  ' if ok else 'FAIL'} {kind} C={C} H={H} K={Kc} R={R} s={st}  {detail}", flush=True)
        failed += 0 if ok else 1
    torch.cuda.synchronize()
    print(f"{failed} failed", flush=True)
    return failed


def timing(B):
    import torch
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
    if len(sys.argv) > 3 and sys.argv[3] == "child":
        what = sys.argv[2]
        rc = 0
        if what in ("parity", "both"):
            rc = parity()
        if what in ("time", "both") and rc == 0:
            timing(int(sys.argv[1]))
        sys.exit(1 if rc else 0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    what = sys.argv[2] if len(sys.argv) > 2 else "both"
    settings = [("base", {"DDL_CONV_BIG": "0"}), ("big", {"DDL_CONV_BIG": "2"}), ("base", {"DDL_CONV_BIG": "0"}),
                ("big", {"DDL_CONV_BIG": "2"})]
    if what == "parity":
        settings = [("big-everything", {"DDL_CONV_BIG": "2"}), ("big-default", {"DDL_CONV_BIG": "1"})]
    rows = []
    for name, e in settings:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(B), what, "child"], env=dict(os.environ, **e),
                           capture_output=True, text=True, timeout=200)
        par = [ln for ln in r.stdout.splitlines() if ln.startswith(("ok", "FAIL")) or ln.endswith("failed")]
        if name != "base" or r.returncode != 0:
            print(f"--- {name} parity (rc {r.returncode})\n" + "\n".join(par), flush=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
        if r.returncode != 0 or (what != "parity" and not line):
            print(f"{name} failed (rc {r.returncode}):\n{r.stdout[-1500:]}\n{r.stderr[-2500:]}")
            sys.exit(1)          # never start another GPU child after a failure
        if line:
            rows.append([float(v) for v in line[0].split()[1:]])
    if rows:
        print(f"B={B}; ms per launch: " + "  ".join(n for n, _ in settings) + "  | big/base (second pair)")
        for i, (name, _, _) in enumerate(CASES):
            print(f"{name:30s} " + " ".join(f"{rows[j][i]:7.4f}" for j in range(len(settings))) + f"  | {rows[3][i] / rows[2][i]:5.2f}")


if __name__ == "__main__":
    main()
