"""Does interleaving two half-batch chains (apply -> conv) on two streams hide BN-apply behind conv?  Forward only.

Per layer: conv(+stats) over both halves -> finalize (needs both) -> apply per half.  apply(h) -> conv_next(h) only
depends on its own half, so apply(h1) can overlap conv_next(h0)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from distributed_deep_learning_amd.ops import kernels as K  # noqa: E402

B = 256
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 56
C0 = int(sys.argv[2]) if len(sys.argv) > 2 else 256
P = C0 // 4
# bottleneck: C0 -> P (1x1) -> P (3x3) -> C0 (1x1), repeated
specs = [(C0, P, 1, 0), (P, P, 3, 1), (P, C0, 1, 0)] * 3
ws = []
for cin, cout, r, pad in specs:
    w = (torch.randn(cout, cin, r, r, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    ws.append(K.filter_repack(w)[0])
gam = [torch.ones(s[1], device=dev) for s in specs]
bet = [torch.zeros(s[1], device=dev) for s in specs]
slabs = [K.new_slab(s[1], dev) for s in specs]
x = torch.randn(B, C0, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


def full():
    h = x
    for i, (cin, cout, r, pad) in enumerate(specs):
        z, st = K.conv2d_fwd(h, ws[i], 1, pad, want_stats=True, slab=slabs[i])
        vec = K.bn_finalize(st, B * H * H, gam[i], bet[i], None, None, 0.1, 1e-5)
        rows = K.vec_rows(vec)
        h, _ = K.bn_apply(z, rows[2], rows[3], relu=True, want_mask=True)
    return h


s1 = torch.cuda.Stream(priority=-1)
NH = 2


def split():
    main = torch.cuda.current_stream()
    hs = list(x.chunk(NH, 0))
    for i, (cin, cout, r, pad) in enumerate(specs):
        zs = []
        # chain 0 on main, chain 1 on s1
        for k in range(NH):
            st_ = main if k == 0 else s1
            with torch.cuda.stream(st_):
                z, st = K.conv2d_fwd(hs[k], ws[i], 1, pad, want_stats=True, slab=slabs[i])
            zs.append(z)
        main.wait_stream(s1)
        vec = K.bn_finalize(slabs[i], B * H * H, gam[i], bet[i], None, None, 0.1, 1e-5)
        rows = K.vec_rows(vec)
        s1.wait_stream(main)
        for k in range(NH):
            st_ = main if k == 0 else s1
            with torch.cuda.stream(st_):
                hs[k], _ = K.bn_apply(zs[k], rows[2], rows[3], relu=True, want_mask=True)
    main.wait_stream(s1)
    return hs


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a = full()
b = torch.cat(split(), 0)
torch.cuda.synchronize()
print("max diff", (a.float() - b.float()).abs().max().item())
print(f"H={H} C={C0}: full-batch chain {timeit(full):.3f} ms | two half-batch chains on two streams {timeit(split):.3f} ms")
