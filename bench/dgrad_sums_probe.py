"""Cross-block link of a bottleneck chain: block k+1's first dgrad (+ masked residual addend) produces dout_k; block k's
last BatchNorm backward needs sum(g), sum(g*z) with g = dout_k * mask_k.
  A: dgrad + addend, then bn_backward (reduce pass + finalize + apply pass)
  B: dgrad + addend with the sums taken in its epilogue (bn_sums), then bn_backward(presummed=True) (finalize + apply)
Parity of B against A (dx of the BN, dgamma, dbeta) and time of both.   python bench/dgrad_sums_probe.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from distributed_deep_learning_amd.ops import kernels as K  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
CL = torch.channels_last


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


print(f"B={B}: wide C, H, planes | A ms | B ms | B/A | max rel err dx / dgamma / dbeta")
for (C, H, P) in [(256, 56, 64), (512, 28, 128), (1024, 14, 256), (2048, 7, 512)]:
    g = torch.Generator(device="cpu").manual_seed(C)
    M = B * H * H
    dz1 = torch.randn(B, P, H, H, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=CL)       # grad of conv1 output
    w = (torch.randn(P, C, 1, 1, generator=g) * 0.05).to(dev).contiguous(memory_format=CL)
    _, wt = K.filter_repack(w)
    dnext = torch.randn(B, C, H, H, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=CL)      # dout of block k+1
    mnext = torch.randint(0, 256, (M, C // 8), generator=g, dtype=torch.uint8).to(dev)                     # its ReLU mask
    z3 = (torch.randn(B, C, H, H, generator=g) * 1.3 + 0.2).to(dev).to(torch.bfloat16).contiguous(memory_format=CL)
    mk = torch.randint(0, 256, (M, C // 8), generator=g, dtype=torch.uint8).to(dev)                        # ReLU mask of block k
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    slab0 = K.bn_stats(z3)
    vec = K.bn_finalize(slab0, M, gamma, torch.zeros(C, device=dev), None, None, 0.1, 1e-5)
    mean, invstd = vec[0], vec[1]
    slabA, slabB = K.new_slab(C, dev), K.new_slab(C, dev)

    def run_a():
        dout = K.conv2d_dgrad(dz1, wt, (H, H), 1, 0, addend=dnext, addend_mask=mnext)
        return K.bn_backward(dout, mk, z3, gamma, mean, invstd, relu=True, slab=slabA)

    def run_b():
        dout = K.conv2d_dgrad(dz1, wt, (H, H), 1, 0, addend=dnext, addend_mask=mnext, bn_sums=(z3, mk, slabB))
        return K.bn_backward(dout, mk, z3, gamma, mean, invstd, relu=True, slab=slabB, presummed=True)

    ra, rb = run_a(), run_b()
    errs = []
    for a, b in ((ra[0], rb[0]), (ra[2], rb[2]), (ra[3], rb[3])):
        errs.append(((a.float() - b.float()).abs().max() / a.float().abs().max().clamp_min(1e-6)).item())
    ta, tb = timeit(run_a), timeit(run_b)
    print(f"  {C:5d} {H:3d} {P:4d} | {ta:.4f} | {tb:.4f} | {tb / ta:.3f} | " + " / ".join(f"{e:.2e}" for e in errs), flush=True)
