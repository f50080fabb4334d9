"""Numerics of every hand-written gfx950 kernel against a plain PyTorch fp32 reference of the same op."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CL = torch.channels_last


def _dev():
    return torch.device("cuda:0")


def _rand_act(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)
    return x.to(_dev()).contiguous(memory_format=CL)


def _close(a, b, rtol=2e-2, atol=2e-2, what=""):
    a, b = a.float(), b.float()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol).sum().item()
    assert bad == 0, f"{what}: {bad}/{a.numel()} mismatches, max err {err.max().item():.4g}, ref max {b.abs().max().item():.4g}"


CONV_CASES = [
    # N, C, H, W, K, R, stride, pad
    (2, 64, 14, 14, 64, 1, 1, 0),
    (2, 64, 14, 14, 256, 1, 1, 0),
    (3, 256, 9, 9, 64, 1, 1, 0),       # M tail (243 rows)
    (2, 128, 16, 16, 128, 3, 1, 1),
    (2, 64, 17, 15, 64, 3, 2, 1),      # odd sizes, stride 2
    (2, 256, 14, 14, 512, 1, 2, 0),    # strided 1x1 (downsample)
    (2, 8, 32, 32, 64, 7, 2, 3),       # stem with padded channels
    (2, 16, 12, 12, 24, 3, 1, 1),      # small channel counts, K tail
    (1, 96, 8, 8, 128, 1, 1, 0),       # C not a multiple of 64
    (2, 32, 10, 10, 72, 5, 1, 2),      # 5x5, OC tail
    (4, 512, 7, 7, 512, 3, 1, 1),
    (2, 64, 56, 56, 64, 3, 1, 1),      # LDS-resident patch kernel: 64-channel tile, wide rows
    (3, 256, 14, 14, 256, 3, 1, 1),    # patch kernel: tiles span image boundaries, 8 channel slices
    (1, 32, 20, 36, 96, 3, 1, 1),      # patch kernel: non-square, OC tail, single slice, M tail
    (4, 8, 28, 28, 8, 5, 1, 2),        # LeNet-5 conv1 (1 -> 6 channels, both padded to 8)
    (4, 8, 14, 14, 16, 5, 1, 0),       # LeNet-5 conv2 (6 -> 16)
    (3, 64, 18, 21, 64, 3, 1, 0),      # unpadded 3x3: output grid smaller than input (filter-gradient cursor wraps)
    (5, 16, 23, 19, 64, 4, 1, 0),      # space-to-depth stem shape, odd sizes (filter rows folded into the column index)
    (3, 16, 115, 115, 64, 4, 1, 0),    # the space-to-depth stem itself (forward: input patch in LDS, filter in registers)
    (2, 16, 37, 51, 64, 4, 1, 0),      # same kernel: 34 output rows = 8 x 4 + 2 (half-empty last stage), 48-pixel rows
    (5, 64, 7, 7, 128, 3, 1, 1),       # halo filter gradient: four 7x7 images per stage, the last stage holds one
    (3, 32, 28, 28, 128, 3, 1, 1),     # halo filter gradient: 8 output rows per stage, the last row block of an image half empty
    (2, 64, 30, 50, 64, 3, 1, 1),      # halo filter gradient: 4 rows of 50 per stage (200 of 224 slots), 30 = 7 x 4 + 2
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case):
    from distributed_deep_learning_amd.ops import kernels as K

    N, C, H, W, Kc, R, stride, pad = case
    x = _rand_act((N, C, H, W), 1)
    w = (torch.randn((Kc, C, R, R), generator=torch.Generator().manual_seed(2)) / (C * R * R) ** 0.5)
    w = w.to(_dev()).contiguous(memory_format=CL)
    w_krsc, w_crsk = K.filter_repack(w)
    wq = w_krsc.permute(0, 3, 1, 2).float()            # the bf16-rounded filter the kernels see
    assert torch.equal(w_crsk.permute(3, 0, 1, 2).float(), wq), "CRSK repack mismatch"

    y, stats = K.conv2d_fwd(x, w_krsc, stride, pad, want_stats=True)
    y_ref = F.conv2d(x.float(), wq, None, stride, pad)
    _close(y, y_ref, what="conv fwd")
    # fused BN partial sums are computed from the rounded bf16 outputs
    s = K.slab_sums(stats)
    yf = y.float()
    _close(s[0], yf.sum((0, 2, 3)), rtol=1e-3, atol=1e-2 * max(1.0, yf.abs().max().item()), what="fused sum")
    _close(s[1], (yf * yf).sum((0, 2, 3)), rtol=1e-3, atol=1e-2, what="fused sumsq")

    dy = _rand_act(tuple(y.shape), 3)
    dx = K.conv2d_dgrad(dy, w_crsk, (H, W), stride, pad)
    dx_ref = torch.nn.grad.conv2d_input((N, C, H, W), wq, dy.float(), stride, pad)
    _close(dx, dx_ref, what="conv dgrad")

    dw = K.conv2d_wgrad(dy, x, R, R, stride, pad)
    dw_ref = torch.nn.grad.conv2d_weight(x.float(), (Kc, C, R, R), dy.float(), stride, pad)
    scale = dw_ref.abs().max().item()
    _close(dw, dw_ref, rtol=1e-2, atol=1e-2 * scale, what="conv wgrad")


@pytest.mark.gpu
def test_stem_forward_kernel_without_statistics_and_its_fallbacks():
    """csrc/conv_stem.hip has one instantiation with and one without the BatchNorm sums; a bias or a ReLU in the epilogue
    is not its business and must take the ring kernels.  All three against fp32."""
    from distributed_deep_learning_amd.ops import kernels as K
    x = _rand_act((3, 16, 35, 35), 5)
    w = (torch.randn((64, 16, 4, 4), generator=torch.Generator().manual_seed(6)) / 16.0).to(_dev()).contiguous(memory_format=CL)
    wk, _ = K.filter_repack(w)
    wq = wk.permute(0, 3, 1, 2).float()
    ref = F.conv2d(x.float(), wq, None, 1, 0)
    y0, st = K.conv2d_fwd(x, wk, 1, 0, want_stats=False)
    assert st is None
    _close(y0, ref, what="stem forward, no statistics")
    y1, _ = K.conv2d_fwd(x, wk, 1, 0, want_stats=True)
    assert torch.equal(y0, y1)
    bias = torch.linspace(-1, 1, 64, device=_dev())
    y2, _ = K.conv2d_fwd(x, wk, 1, 0, bias=bias, relu=True)
    _close(y2, F.relu(ref + bias.view(1, -1, 1, 1)), what="stem shape with bias + ReLU (ring kernels)")


@pytest.mark.parametrize("case", [(2, 64, 14, 14, 256, 1, 1, 0), (3, 256, 9, 9, 64, 1, 1, 0), (2, 128, 16, 16, 128, 3, 1, 1),
                                  (2, 64, 16, 16, 64, 3, 2, 1)])
@pytest.mark.parametrize("masked", [False, True])
def test_conv_dgrad_addend_epilogue(case, masked, monkeypatch):
    """dx = dgrad(dy) + addend [* relu-mask bits]: the fused residual-gradient accumulation (both conv kernels)."""
    from distributed_deep_learning_amd.ops import kernels as K

    N, C, H, W, Kc, R, stride, pad = case
    w = (torch.randn((Kc, C, R, R), generator=torch.Generator().manual_seed(2)) / (C * R * R) ** 0.5)
    w = w.to(_dev()).contiguous(memory_format=CL)
    w_krsc, w_crsk = K.filter_repack(w)
    wq = w_krsc.permute(0, 3, 1, 2).float()
    P, Q = K.conv_out_size(H, R, stride, pad), K.conv_out_size(W, R, stride, pad)
    dy = _rand_act((N, Kc, P, Q), 3)
    addend = _rand_act((N, C, H, W), 4)
    mask = None
    keep = torch.ones((N, C, H, W), device=_dev())
    if masked:
        # the mask in BN-apply layout: produced by the real kernel from a random pre-activation
        pre = _rand_act((N, C, H, W), 5)
        one, zero = torch.ones(C, device=_dev()), torch.zeros(C, device=_dev())
        yact, mask = K.bn_apply(pre, one, zero, relu=True, want_mask=True)
        keep = (yact.float() > 0).float()
    dx = K.conv2d_dgrad(dy, w_crsk, (H, W), stride, pad, addend=addend, addend_mask=mask)
    ref = torch.nn.grad.conv2d_input((N, C, H, W), wq, dy.float(), stride, pad) + addend.float() * keep
    _close(dx, ref, what="dgrad + addend")


def test_conv_dgrad_compact_strided_addend():
    """addend_step=2: a compact [N,C,ceil(H/2),ceil(W/2)] addend lands on the even pixels only (projection shortcut)."""
    from distributed_deep_learning_amd.ops import kernels as K

    for (N, C, H, W, Kc) in [(2, 256, 14, 14, 64), (3, 64, 9, 11, 128)]:
        w = (torch.randn((Kc, C, 1, 1), generator=torch.Generator().manual_seed(2)) / C ** 0.5)
        w = w.to(_dev()).contiguous(memory_format=CL)
        w_krsc, w_crsk = K.filter_repack(w)
        dy = _rand_act((N, Kc, H, W), 3)
        comp = _rand_act((N, C, (H + 1) // 2, (W + 1) // 2), 4)
        dx = K.conv2d_dgrad(dy, w_crsk, (H, W), 1, 0, addend=comp, addend_step=2)
        ref = torch.nn.grad.conv2d_input((N, C, H, W), w_krsc.permute(0, 3, 1, 2).float(), dy.float(), 1, 0)
        ref[:, :, ::2, ::2] += comp.float()
        _close(dx, ref, what="dgrad + compact addend")


def test_conv_bias_relu_epilogue():
    from distributed_deep_learning_amd.ops import kernels as K

    x = _rand_act((2, 64, 12, 12), 5)
    w = torch.randn((128, 64, 3, 3), generator=torch.Generator().manual_seed(6)) / 24.0
    w = w.to(_dev()).contiguous(memory_format=CL)
    b = torch.randn(128, generator=torch.Generator().manual_seed(7)).to(_dev())
    w_krsc, _ = K.filter_repack(w)
    y, _ = K.conv2d_fwd(x, w_krsc, 1, 1, bias=b, relu=True)
    ref = F.relu(F.conv2d(x.float(), w_krsc.permute(0, 3, 1, 2).float(), b, 1, 1))
    _close(y, ref, what="conv+bias+relu")


@pytest.mark.parametrize("shape", [(4, 64, 14, 14), (2, 96, 7, 9), (3, 256, 5, 5), (2, 2048, 4, 4), (2, 4096, 2, 2)])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False)])
def test_bn_train_fwd_bwd(shape, relu, res):
    from distributed_deep_learning_amd.ops import kernels as K

    N, C, H, W = shape
    x = _rand_act(shape, 11, scale=2.0) + 0.5
    r = _rand_act(shape, 12) if res else None
    g = torch.Generator().manual_seed(13)
    gamma = (torch.rand(C, generator=g) + 0.5).to(_dev())
    beta = (torch.randn(C, generator=g) * 0.1).to(_dev())
    rm, rv = torch.zeros(C, device=_dev()), torch.ones(C, device=_dev())
    eps, mom = 1e-5, 0.1

    slab = K.bn_stats(x)
    vec = K.bn_finalize(slab, N * H * W, gamma, beta, rm, rv, mom, eps)
    y, mask = K.bn_apply(x, vec[2], vec[3], residual=r, relu=relu, want_mask=True)
    if relu:
        bits = ((mask.long().unsqueeze(-1) >> torch.arange(8, device=mask.device)) & 1).reshape(N, H, W, C).permute(0, 3, 1, 2)
        assert torch.equal(bits.bool(), y.float() > 0), "ReLU bit mask mismatch"

    xf = x.float().requires_grad_(True)
    rf = r.float().requires_grad_(True) if res else None
    gp, bp = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm2, rv2 = torch.zeros(C, device=_dev()), torch.ones(C, device=_dev())
    ref = F.batch_norm(xf, rm2, rv2, gp, bp, True, mom, eps)
    if res:
        ref = ref + rf
    if relu:
        ref = F.relu(ref)
    _close(y, ref, what="bn fwd")
    _close(rm, rm2, rtol=1e-3, atol=1e-4, what="running_mean")
    _close(rv, rv2, rtol=1e-3, atol=1e-4, what="running_var")

    dy = _rand_act(shape, 14)
    # use the kernel's own y for the ReLU mask so rounding at 0 cannot flip elements between impls
    dx, dres, dgamma, dbeta = K.bn_backward(dy, mask, x, gamma, vec[0], vec[1], relu=relu, want_dres=res)
    mask = (y.float() > 0).float() if relu else torch.ones_like(y, dtype=torch.float32)
    # reference backward with the same mask: d(pre-relu) = dy * mask
    pre = F.batch_norm(xf, None, None, gp, bp, True, mom, eps)
    pre.backward(dy.float() * mask)
    _close(dx, xf.grad, rtol=3e-2, atol=3e-2, what="bn dx")
    scale = max(1.0, gp.grad.abs().max().item())
    _close(dgamma, gp.grad, rtol=1e-2, atol=1e-2 * scale, what="dgamma")
    _close(dbeta, bp.grad, rtol=1e-2, atol=1e-2 * scale, what="dbeta")
    if res:
        _close(dres, dy.float() * mask, what="dres")


@pytest.mark.parametrize("k,stride,pad,shape", [(3, 2, 1, (2, 64, 16, 16)), (2, 2, 0, (2, 16, 8, 8)), (3, 2, 1, (1, 8, 7, 9))])
def test_maxpool(k, stride, pad, shape):
    from distributed_deep_learning_amd.ops import kernels as K

    x = _rand_act(shape, 21)
    x = torch.relu(x)  # plenty of ties at zero: exercises the first-max rule
    y, idx = K.maxpool_fwd(x, k, stride, pad)
    xf = x.float().requires_grad_(True)
    ref = F.max_pool2d(xf, k, stride, pad)
    assert torch.equal(y.float(), ref)
    dy = _rand_act(tuple(y.shape), 22)
    dx = K.maxpool_bwd(dy, idx, shape[2:], k, stride, pad)
    ref.backward(dy.float())
    _close(dx, xf.grad, what="maxpool bwd")


def test_avgpool_and_gap():
    from distributed_deep_learning_amd.ops import kernels as K

    x = _rand_act((2, 64, 8, 8), 31)
    y = K.avgpool_fwd(x, 2, 2)
    _close(y, F.avg_pool2d(x.float(), 2, 2), what="avgpool fwd")
    dy = _rand_act(tuple(y.shape), 32)
    dx = K.avgpool_bwd(dy, (8, 8), 2, 2)
    xf = x.float().requires_grad_(True)
    F.avg_pool2d(xf, 2, 2).backward(dy.float())
    _close(dx, xf.grad, what="avgpool bwd")

    x = _rand_act((3, 2048, 7, 7), 33)
    g = K.gap_fwd(x)
    _close(g, x.float().mean((2, 3)), what="gap fwd")
    dg = torch.randn((3, 2048), device=_dev()).to(torch.bfloat16)
    dx = K.gap_bwd(dg, (7, 7))
    _close(dx, (dg.float() / 49.0)[:, :, None, None].expand(3, 2048, 7, 7), what="gap bwd")
    x = _rand_act((2, 96, 5, 5), 34)
    _close(K.gap_fwd(x), x.float().mean((2, 3)), what="gap fwd c96")


def test_eltwise():
    from distributed_deep_learning_amd.ops import kernels as K

    a, b = _rand_act((2, 64, 6, 6), 41), _rand_act((2, 64, 6, 6), 42)
    _close(K.add(a, b, relu=True), F.relu(a.float() + b.float()), what="add_relu")
    y = K.relu_fwd(a)
    assert torch.equal(y.float(), F.relu(a.float()))
    _close(K.relu_bwd(b, y), b.float() * (y.float() > 0), what="relu_bwd")
    _close(K.bias_grad(a), a.float().sum((0, 2, 3)), rtol=1e-3, atol=1e-2, what="bias_grad")
    img = torch.randn((2, 3, 10, 10), device=_dev())
    p = K.nchw_to_nhwc_pad(img, 8)
    assert p.shape == (2, 8, 10, 10) and p.is_contiguous(memory_format=CL)
    _close(p[:, :3], img, what="nchw->nhwc")
    assert p[:, 3:].abs().max().item() == 0
    f = torch.randn(4096, device=_dev())
    assert torch.equal(K.cast_f32_bf16(f), f.to(torch.bfloat16))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,V", [(32, 1000), (7, 10), (5, 6)])
def test_softmax_xent(dtype, B, V):
    from distributed_deep_learning_amd.ops import kernels as K

    g = torch.Generator().manual_seed(51)
    z = (torch.randn((B, V), generator=g) * 3).to(dtype).to(_dev())
    lab = torch.randint(0, V, (B,), generator=g).to(_dev())
    out, dz = K.softmax_xent(z, lab)
    zf = z.float().requires_grad_(True)
    ref = F.cross_entropy(zf, lab)
    ref.backward()
    assert abs(out[0].item() - ref.item()) < 2e-3 * max(1.0, abs(ref.item()))
    assert out[1].item() == (zf.argmax(1) == lab).sum().item()
    _close(dz, zf.grad, rtol=2e-2, atol=1e-4, what="xent grad")
    # probability (one-hot float) targets: the reference repo's target format
    probs = F.one_hot(lab, V).float()
    out2, dz2 = K.softmax_xent(z, probs)
    assert abs(out2[0].item() - ref.item()) < 2e-3 * max(1.0, abs(ref.item()))
    _close(dz2, zf.grad, rtol=2e-2, atol=1e-4, what="xent grad (prob targets)")


@pytest.mark.parametrize("shape", [(4, 64, 14, 14), (2, 512, 7, 9), (3, 4096, 2, 2)])
def test_relu_bwd_bias_grad_fused(shape):
    from distributed_deep_learning_amd.ops import kernels as K

    dy, y = _rand_act(shape, 11), _rand_act(shape, 12)
    dx, db = K.relu_bwd_bias_grad(dy, y)
    ref = dy.float() * (y.float() > 0)
    assert torch.equal(dx.float(), ref)
    _close(db, ref.sum((0, 2, 3)), rtol=1e-3, atol=1e-2, what="bias grad")
    acc = torch.ones(shape[1], device=_dev())
    dx2, out = K.relu_bwd_bias_grad(dy, y, grad_out=acc)
    assert out is acc and torch.equal(dx2, dx)
    _close(acc, 1.0 + ref.sum((0, 2, 3)), rtol=1e-3, atol=1e-2, what="bias grad accumulate")


@pytest.mark.parametrize("pooled_sums", [False, True], ids=["gather-sums", "pooled-sums"])
@pytest.mark.parametrize("shape,pool", [((2, 64, 16, 16), (3, 2, 1)), ((3, 32, 9, 11), (3, 2, 1)), ((2, 16, 8, 8), (2, 2, 0)),
                                        ((4, 64, 28, 30), (3, 2, 1)),       # even 3/2/1 cases: the 2x2-quad apply kernel
                                        ((3, 64, 20, 24), (3, 2, 1))])      # W % 4 == 0: the paired forward kernel too
def test_bn_relu_maxpool_fused_unit(shape, pool, pooled_sums):
    """maxpool(relu(bn(x))) forward + its BN backward from the POOLED gradient vs autograd on fp32 torch ops; the two
    backward sums taken by gathering per full-resolution pixel, or over the pooled tensors (zmax / pmask of the forward)."""
    from distributed_deep_learning_amd.ops import kernels as K

    N, C, H, W = shape
    k, s, p = pool
    x = _rand_act(shape, 21, scale=1.5)
    gamma = (torch.rand(C, generator=torch.Generator().manual_seed(22)) + 0.5).to(_dev())
    beta = (torch.randn(C, generator=torch.Generator().manual_seed(23)) * 0.3).to(_dev())
    slab = K.bn_stats(x)
    vec = K.bn_finalize(slab, N * H * W, gamma, beta, None, None, 0.1, 1e-5)
    aux = None
    if pooled_sums:
        y, idx, aux = K.bn_relu_maxpool_fwd(x, vec[2], vec[3], k, s, p, want_aux=True)
    else:
        y, idx = K.bn_relu_maxpool_fwd(x, vec[2], vec[3], k, s, p)

    xr = x.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.max_pool2d(F.relu(F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)), k, s, p)
    _close(y, ref, what="fused bn+relu+maxpool fwd")
    dp = _rand_act(tuple(y.shape), 24)
    ref.backward(dp.float())
    dx, dg, db = K.bn_relu_maxpool_bwd(dp, idx, x, gamma, vec[0], vec[1], k, s, p, aux=aux)
    _close(dg, gr.grad, rtol=2e-2, atol=2e-2 * max(1.0, gr.grad.abs().max().item()), what="dgamma")
    _close(db, br.grad, rtol=2e-2, atol=2e-2 * max(1.0, br.grad.abs().max().item()), what="dbeta")
    # bf16 ties in the pooled max can route a gradient to a different pixel of the window: compare in aggregate
    err = (dx.float() - xr.grad).abs()
    assert (err > 3e-2 + 3e-2 * xr.grad.abs()).float().mean().item() < 0.01, err.max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,H,K,R,stride,pad", [(32, 64, 56, 64, 3, 1, 1), (16, 16, 115, 64, 4, 1, 0), (24, 512, 7, 512, 3, 1, 1),
                                                  (32, 256, 14, 1024, 1, 1, 0), (20, 128, 28, 128, 3, 2, 1)])
def test_wgrad_is_bit_reproducible_and_accumulates(N, C, H, K, R, stride, pad):
    """Deterministic mode (``ops.state.set_deterministic`` / ``--deterministic`` / ``torch.use_deterministic_algorithms``):
    the split-M partial tiles of a filter gradient meet in slabs + an ordered reduce (csrc/conv_wgrad.hip), not in
    float atomics: two launches on the same operands agree to the BIT, for every tile family (64-channel 3x3 with
    hundreds of splits, the folded stem tile, wide late-layer tiles, plain 1x1, strided), and ``out=`` is accumulated
    into.  The default (atomics) gives the same numbers up to summation order."""
    from distributed_deep_learning_amd.ops import kernels as Kk
    from distributed_deep_learning_amd.ops import state
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn((N, C, H, H), generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    P = (H + 2 * pad - R) // stride + 1
    dy = torch.randn((N, K, P, P), generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    fast = Kk.conv2d_wgrad(dy, x, R, R, stride, pad)
    state.set_deterministic(True)
    try:
        assert state.deterministic()
        a = Kk.conv2d_wgrad(dy, x, R, R, stride, pad)
        b = Kk.conv2d_wgrad(dy, x, R, R, stride, pad)
        out = torch.ones((K, R, R, C), dtype=torch.float32, device="cuda").permute(0, 3, 1, 2)
        Kk.conv2d_wgrad(dy, x, R, R, stride, pad, out=out)
    finally:
        state.set_deterministic(False)
    assert torch.equal(a, b)
    ref = torch.nn.grad.conv2d_weight(x.float(), (K, C, R, R), dy.float(), stride, pad)
    assert float((a - ref).abs().max()) < 2e-2 * float(ref.abs().max()) + 1e-3
    assert float((out - (a + 1.0)).abs().max()) < 1e-5 * float(a.abs().max()) + 1e-6
    assert float((fast - a).abs().max()) < 1e-4 * float(a.abs().max()) + 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W", [(3, 56, 56), (2, 28, 28), (5, 32, 32), (2, 16, 16), (3, 20, 112), (80, 56, 56), (300, 32, 32)])
def test_conv3x3_c64_kernel_matches_fp32_and_the_ring_kernels(N, H, W):
    """csrc/conv_c64.hip (filter + input patch in LDS) is chosen from ~150 images at 56 x 56 on; the threshold is lifted
    here.  Forward with and without BatchNorm sums, data gradient, ragged last row blocks, three to four stages per
    block in the two large cases (the counted wait that leaves a stage's stores in flight), against fp32 and against the
    ring kernels' bits (same products, same fp32 accumulation order up to the MFMA's own: compared to 1e-2 of a bf16 ulp's
    worth of the output scale)."""
    from distributed_deep_learning_amd.ops import kernels as K
    x = _rand_act((N, 64, H, W), 31)
    w = (torch.randn((64, 64, 3, 3), generator=torch.Generator().manual_seed(32)) / 24.0).to(_dev()).contiguous(memory_format=CL)
    wk, wc = K.filter_repack(w)
    wq = wk.permute(0, 3, 1, 2).float()
    dy = _rand_act((N, 64, H, W), 33)
    y_ring, st_ring = K.conv2d_fwd(x, wk, 1, 1, want_stats=True)
    dx_ring = K.conv2d_dgrad(dy, wc, (H, W), 1, 1)
    from distributed_deep_learning_amd.ops import _lib
    launches = _lib.require().ddl_conv_c64_launches()
    K.set_conv_c64_min_stages(0)
    try:
        y, st = K.conv2d_fwd(x, wk, 1, 1, want_stats=True)
        y0, _ = K.conv2d_fwd(x, wk, 1, 1, want_stats=False)
        dx = K.conv2d_dgrad(dy, wc, (H, W), 1, 1)
    finally:
        K.set_conv_c64_min_stages()
    assert _lib.require().ddl_conv_c64_launches() - launches == (3 if W != 112 else 0)   # 112-pixel rows: patch too big, ring kernels
    ref = F.conv2d(x.float(), wq, None, 1, 1)
    _close(y, ref, what="c64 forward")
    assert torch.equal(y, y0)
    assert float((y.float() - y_ring.float()).abs().max()) <= 0.02 * float(ref.abs().max())
    s, s_ring, yf = K.slab_sums(st), K.slab_sums(st_ring), y.float()
    _close(s[0], yf.sum((0, 2, 3)), rtol=1e-3, atol=1e-2 * max(1.0, yf.abs().max().item()), what="c64 fused sum")
    _close(s[1], (yf * yf).sum((0, 2, 3)), rtol=1e-3, atol=1e-2, what="c64 fused sumsq")
    _close(s[1], s_ring[1], rtol=2e-3, atol=1e-2, what="c64 vs ring sumsq")
    dref = torch.nn.grad.conv2d_input((N, 64, H, W), wq, dy.float(), 1, 1)
    _close(dx, dref, what="c64 dgrad")
    assert float((dx.float() - dx_ring.float()).abs().max()) <= 0.02 * float(dref.abs().max())


@pytest.mark.gpu
def test_late_round3_kernels_run_at_their_shipped_thresholds():
    """The other tests lift the size thresholds of the halo filter gradient and of the 64-channel 3x3 kernel; this one
    leaves them alone and uses operands just big enough for the shipped selection (320 images of 256 x 14 x 14: 20 stages
    per halo block; 160 images of 64 x 56 x 56: 1120 stages of the 64-channel kernel), checks through the launch counters
    that those kernels ran, and compares with fp32."""
    from distributed_deep_learning_amd.ops import _lib
    from distributed_deep_learning_amd.ops import kernels as K
    lib = _lib.require()
    x = _rand_act((320, 256, 14, 14), 51)
    dy = _rand_act((320, 256, 14, 14), 52)
    h0 = lib.ddl_wgrad_halo_launches()
    dw = K.conv2d_wgrad(dy, x, 3, 3, 1, 1)
    assert lib.ddl_wgrad_halo_launches() == h0 + 1
    ref = torch.nn.grad.conv2d_weight(x.float(), (256, 256, 3, 3), dy.float(), 1, 1)
    _close(dw, ref, rtol=1e-2, atol=1e-2 * ref.abs().max().item(), what="halo wgrad at the shipped threshold")
    del x, dy, dw, ref
    x = _rand_act((160, 64, 56, 56), 53)
    w = (torch.randn((64, 64, 3, 3), generator=torch.Generator().manual_seed(54)) / 24.0).to(_dev()).contiguous(memory_format=CL)
    wk, wc = K.filter_repack(w)
    wq = wk.permute(0, 3, 1, 2).float()
    c0 = lib.ddl_conv_c64_launches()
    y, st = K.conv2d_fwd(x, wk, 1, 1, want_stats=True)
    dx = K.conv2d_dgrad(x, wc, (56, 56), 1, 1)
    assert lib.ddl_conv_c64_launches() == c0 + 2
    ref = F.conv2d(x[:8].float(), wq, None, 1, 1)
    _close(y[:8], ref, what="c64 forward at the shipped threshold")
    _close(y[-8:], F.conv2d(x[-8:].float(), wq, None, 1, 1), what="c64 forward, last images")
    yf = y.float()
    _close(K.slab_sums(st)[0], yf.sum((0, 2, 3)), rtol=1e-3, atol=1e-2 * max(1.0, yf.abs().max().item()), what="c64 fused sum")
    dref = torch.nn.grad.conv2d_input((8, 64, 56, 56), wq, x[:8].float(), 1, 1)
    _close(dx[:8], dref, what="c64 dgrad at the shipped threshold")


HALO_CASES = [
    # N, C, H, W, K, R, pad: what csrc/conv_wgrad.hip's halo kernel is built for, at sizes a test can afford
    (3, 256, 14, 14, 256, 3, 1),       # one image per stage, 128-filter tiles, 8 channel slices
    (5, 64, 7, 7, 128, 3, 1),          # four 7x7 images per stage, the last stage holds one
    (3, 32, 28, 28, 128, 3, 1),        # 8 output rows per stage, the last row block of an image half empty
    (2, 64, 30, 50, 64, 3, 1),         # 64-filter tiles, 4 rows of 50 per stage (200 of 224 slots), 30 = 7 x 4 + 2
    (2, 64, 56, 56, 64, 3, 1),         # ResNet-50 layer 1
    (3, 64, 18, 21, 64, 3, 0),         # unpadded
    (5, 16, 23, 19, 64, 4, 0),         # space-to-depth stem shape, odd sizes: one filter row per wave, three stages in LDS
    (2, 16, 115, 115, 64, 4, 0),       # the stem itself
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", HALO_CASES)
@pytest.mark.parametrize("deterministic", [False, True])
def test_halo_filter_gradient_matches_fp32(case, deterministic):
    """The halo filter-gradient kernel is only chosen where a block gets >= 20 stages (batch sizes no test can afford on
    every shape), so the threshold is lifted for the test: small batches, ragged last stages, both meeting modes."""
    from distributed_deep_learning_amd.ops import kernels as K
    from distributed_deep_learning_amd.ops import state
    N, C, H, W, Kc, R, pad = case
    x = _rand_act((N, C, H, W), 11)
    P, Q = H + 2 * pad - R + 1, W + 2 * pad - R + 1
    dy = _rand_act((N, Kc, P, Q), 12)
    ref = torch.nn.grad.conv2d_weight(x.float(), (Kc, C, R, R), dy.float(), 1, pad)
    ring = K.conv2d_wgrad(dy, x, R, R, 1, pad)
    from distributed_deep_learning_amd.ops import _lib
    launches = _lib.require().ddl_wgrad_halo_launches()
    K.set_wgrad_halo_min_stages(0)
    state.set_deterministic(deterministic)
    try:
        dw = K.conv2d_wgrad(dy, x, R, R, 1, pad)
        assert _lib.require().ddl_wgrad_halo_launches() == launches + 1
        again = K.conv2d_wgrad(dy, x, R, R, 1, pad)
        acc = torch.ones((Kc, R, R, C), dtype=torch.float32, device=_dev()).permute(0, 3, 1, 2)
        K.conv2d_wgrad(dy, x, R, R, 1, pad, out=acc)
    finally:
        state.set_deterministic(False)
        K.set_wgrad_halo_min_stages()
    scale = ref.abs().max().item()
    _close(dw, ref, rtol=1e-2, atol=1e-2 * scale, what="halo wgrad")
    _close(dw, ring, rtol=1e-4, atol=1e-4 * scale, what="halo vs ring kernels")
    _close(acc, dw + 1.0, rtol=1e-5, atol=1e-5 * scale + 1e-6, what="accumulates into out")
    if deterministic:
        assert torch.equal(dw, again)


# ----------------------------------------------------------------------------------------------- fused LSTM layer
@pytest.mark.gpu
@pytest.mark.parametrize("B,T,I,H", [(40, 64, 32, 128), (16, 7, 64, 64), (33, 12, 128, 128), (5, 20, 32, 64)])
def test_lstm_layer_matches_fp32_reference(B, T, I, H):
    """Fused LSTM layer (csrc/lstm.hip) vs torch.nn.LSTM in fp32 on the CPU: outputs and all gradients."""
    from distributed_deep_learning_amd import nn as dnn
    torch.manual_seed(7)
    ref = torch.nn.LSTM(I, H, batch_first=True)
    ours = dnn.LSTM(I, H).cuda()
    ours.load_state_dict(ref.state_dict())
    x = torch.randn(B, T, I)
    w = torch.randn(B, T, H) / T ** 0.5           # every time step contributes to the loss
    xr = x.clone().requires_grad_(True)
    out_r, (h_r, _) = ref(xr)
    ((out_r * w).sum() + h_r.sum()).backward()
    xg = x.cuda().requires_grad_(True)
    assert ours.native_ok(xg)
    out_g, (h_g, _) = ours(xg)
    assert out_g.dtype == torch.bfloat16 and out_g.shape == (B, T, H)
    ((out_g.float() * w.cuda()).sum() + h_g.float().sum()).backward()
    torch.cuda.synchronize()

    def rel(a, b):
        return float((a.float().cpu() - b).norm() / (b.norm() + 1e-12))

    assert rel(out_g, out_r) < 2e-2
    assert rel(h_g[0], h_r[0]) < 2e-2
    assert rel(xg.grad, xr.grad) < 3e-2
    for name in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
        assert rel(getattr(ours, name).grad, getattr(ref, name).grad) < 3e-2, name


@pytest.mark.gpu
@pytest.mark.parametrize("M,Kr,N", [(100, 256, 32), (4096 + 37, 512, 128), (31, 512, 64), (128, 256, 64)])
def test_rows_gemm_nt_matches_fp32(M, Kr, N):
    """The LSTM input-gradient GEMM (csrc/lstm.hip: rows_gemm_nt_kernel), ragged row counts included."""
    from distributed_deep_learning_amd.ops import kernels as K
    g = torch.Generator().manual_seed(M)
    a = torch.randn((M, Kr), generator=g).to(torch.bfloat16).cuda()
    bt = (torch.randn((N, Kr), generator=g) / Kr ** 0.5).to(torch.bfloat16).cuda()
    out = K.rows_gemm_nt(a, bt)
    ref = a.float() @ bt.float().t()
    assert out.shape == (M, N) and out.dtype == torch.bfloat16
    assert float((out.float() - ref).abs().max()) < 1e-2 * float(ref.abs().max()) + 1e-3


@pytest.mark.gpu
def test_lstm_weight_gradients_go_straight_into_the_arena():
    """With the parameters in a ParamArena the layer reads the arena's bf16 shadow pairs and accumulates both weight
    gradients in place on the side stream; results equal the private-copy path."""
    from distributed_deep_learning_amd import nn as dnn
    from distributed_deep_learning_amd.ops import streams
    from distributed_deep_learning_amd.parallel import ParamArena
    torch.manual_seed(4)
    plain = dnn.LSTM(32, 128).cuda()
    pooled = dnn.LSTM(32, 128).cuda()
    pooled.load_state_dict(plain.state_dict())
    arena = ParamArena(pooled.named_parameters())
    assert pooled.weight_ih_l0._ddl_shadow_t.shape == (32, 512) and pooled.weight_hh_l0._ddl_shadow_t.shape == (128, 512)
    x = torch.randn(48, 64, 32, device="cuda")
    w = torch.randn(48, 64, 128, device="cuda") / 8.0
    grads = {}
    for tag, m in (("plain", plain), ("arena", pooled)):
        xg = x.clone().requires_grad_(True)
        out, _ = m(xg)
        (out.float() * w).sum().backward()
        streams.join()
        torch.cuda.synchronize()
        grads[tag] = [xg.grad.float()] + [p.grad.float().clone() for p in m.parameters()]
    assert arena.grad_is_arena_view(pooled.weight_ih_l0) and arena.grad_is_arena_view(pooled.weight_hh_l0)
    for a, b in zip(grads["plain"], grads["arena"]):
        assert float((a - b).norm()) <= 2e-3 * float(a.norm()) + 1e-6


@pytest.mark.gpu
def test_lstm_layer_no_grad_and_fallbacks():
    from distributed_deep_learning_amd import nn as dnn
    torch.manual_seed(3)
    m = dnn.LSTM(32, 128).cuda()
    x = torch.randn(24, 10, 32, device="cuda")
    with torch.no_grad():
        out, (h, c) = m(x)                          # inference: nothing saved
    ref_out, (ref_h, _) = torch.nn.LSTM.forward(m, x)
    assert float((out.float() - ref_out).abs().max()) < 3e-2
    assert c is None and h.shape == (1, 24, 128)
    odd = dnn.LSTM(20, 48).cuda()                   # no compiled instance -> stock path, fp32 out
    o2, _ = odd(torch.randn(4, 6, 20, device="cuda"))
    assert o2.dtype == torch.float32 and not odd.native_ok(torch.empty(4, 6, 20, device="cuda"))


# ----------------------------------------------------------------------------------------------- space-to-depth stem
@pytest.mark.gpu
def test_s2d_stem_matches_direct_7x7_stem():
    """7x7/2 stem as space-to-depth input + regrouped 4x4/1 filter == the direct stem: activations, filter / BN grads."""
    from distributed_deep_learning_amd import nn as dnn
    from distributed_deep_learning_amd.nn import modules as M
    torch.manual_seed(5)
    x = torch.randn(6, 3, 64, 48, device="cuda")
    gout = None
    res = {}
    for tag, flag in (("direct", False), ("s2d", True)):
        torch.manual_seed(9)
        conv = dnn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
        bn = dnn.BatchNorm2d(64).cuda()
        pool = dnn.MaxPool2d(3, 2, 1)
        M._S2D_STEM = flag
        try:
            y = M.conv_bn_relu_pool(conv, bn, pool, x)
        finally:
            M._S2D_STEM = True
        if gout is None:
            gout = torch.randn_like(y.float()).to(y.dtype)
        y.backward(gout)
        res[tag] = (y.float(), conv.weight.grad.float().clone(), bn.weight.grad.float().clone(), bn.bias.grad.float().clone())
    # fp32 reference of the same unit
    torch.manual_seed(9)
    conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
    bn = torch.nn.BatchNorm2d(64).cuda()
    yr = torch.nn.functional.max_pool2d(torch.relu(bn(conv(x))), 3, 2, 1)
    yr.backward(gout.float())

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-12))

    assert res["s2d"][0].shape == yr.shape
    assert rel(res["s2d"][0], yr.detach()) < 2e-2
    # same bf16 operands, same fp32 accumulation: the two formulations agree to rounding of the summation order
    assert rel(res["s2d"][0], res["direct"][0]) < 1e-2 and rel(res["s2d"][1], res["direct"][1]) < 1e-3
    # against fp32 the filter gradient carries the bf16 noise of the BN-backward input (8 % on this tiny sample,
    # uniform over the 49 taps, identical for both formulations)
    assert rel(res["s2d"][1], conv.weight.grad) < 0.15
    assert rel(res["s2d"][2], bn.weight.grad) < 5e-2 and rel(res["s2d"][3], bn.bias.grad) < 5e-2


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 64, 14, 14), (3, 256, 9, 7), (2, 24, 5, 5)])
def test_bn_backward_pair_equals_two_single_backwards(shape):
    """Paired BN backward (one read of dy + mask for a block's last BN and its shortcut BN) == two bn_backward calls,
    both with fresh outputs and accumulating into existing gradient vectors."""
    from distributed_deep_learning_amd.ops import kernels as K
    N, C, H, W = shape
    xa = _rand_act(shape, 21, scale=2.0) + 0.5
    xb = _rand_act(shape, 22, scale=0.7) - 0.2
    g = torch.Generator().manual_seed(23)
    stats = []
    for x in (xa, xb):
        gamma = (torch.rand(C, generator=g) + 0.5).to(_dev())
        beta = (torch.randn(C, generator=g) * 0.1).to(_dev())
        vec = K.bn_finalize(K.bn_stats(x), N * H * W, gamma, beta, torch.zeros(C, device=_dev()), torch.ones(C, device=_dev()),
                            0.1, 1e-5)
        stats.append((gamma, vec))
    (ga, va), (gb, vb) = stats
    y, mask = K.bn_apply(xa, va[2], va[3], residual=xb, relu=True, want_mask=True, res_scale=vb[2], res_shift=vb[3])
    ref_y = torch.relu(xa.float() * va[2].view(1, -1, 1, 1) + va[3].view(1, -1, 1, 1)
                       + xb.float() * vb[2].view(1, -1, 1, 1) + vb[3].view(1, -1, 1, 1))
    _close(y, ref_y, what="dual-affine apply")
    dy = _rand_act(shape, 24)
    one_a = K.bn_backward(dy, mask, xa, ga, va[0], va[1], relu=True)
    one_b = K.bn_backward(dy, mask, xb, gb, vb[0], vb[1], relu=True)
    dxa, dxb, gra, grb = K.bn_backward_pair(dy, mask, xa, xb, (ga, va[0], va[1]), (gb, vb[0], vb[1]))
    assert torch.equal(dxa, one_a[0]) or (dxa.float() - one_a[0].float()).abs().max() < 2e-2
    assert torch.equal(dxb, one_b[0]) or (dxb.float() - one_b[0].float()).abs().max() < 2e-2
    for got, want in ((gra[0], one_a[2]), (gra[1], one_a[3]), (grb[0], one_b[2]), (grb[1], one_b[3])):
        _close(got, want, rtol=1e-4, atol=1e-3 * max(1.0, float(want.abs().max())), what="pair dgamma/dbeta")
    # accumulate mode: adds to what is already there
    acc = [torch.full((C,), 2.0, device=_dev()) for _ in range(4)]
    K.bn_backward_pair(dy, mask, xa, xb, (ga, va[0], va[1]), (gb, vb[0], vb[1]), grad_outs=((acc[0], acc[1]), (acc[2], acc[3])))
    for got, want in zip(acc, (one_a[2], one_a[3], one_b[2], one_b[3])):
        _close(got - 2.0, want, rtol=1e-4, atol=1e-3 * max(1.0, float(want.abs().max())), what="pair accumulate")


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # N, C(in of the conv = channels of dx), H, W, K, R, stride, pad, addend
    (4, 64, 14, 14, 256, 1, 1, 0, False),
    (3, 128, 12, 10, 128, 3, 1, 1, False),
    (2, 64, 16, 16, 64, 3, 2, 1, False),       # strided: one launch per parity class, sums accumulate across them
    (2, 256, 9, 9, 64, 1, 1, 0, True),         # with a masked addend: sums are of the final gradient
])
def test_dgrad_epilogue_bn_backward_sums(case):
    """conv2d_dgrad(bn_sums=...) + bn_backward(presummed=True) == conv2d_dgrad + full bn_backward."""
    from distributed_deep_learning_amd.ops import kernels as K
    N, C, H, W, Kc, R, stride, pad, with_addend = case
    P, Q = K.conv_out_size(H, R, stride, pad), K.conv_out_size(W, R, stride, pad)
    z = _rand_act((N, C, H, W), 31, scale=1.5) + 0.3            # input of the BN whose activation feeds the conv
    g = torch.Generator().manual_seed(32)
    gamma = (torch.rand(C, generator=g) + 0.5).to(_dev())
    beta = (torch.randn(C, generator=g) * 0.1).to(_dev())
    vec = K.bn_finalize(K.bn_stats(z), N * H * W, gamma, beta, torch.zeros(C, device=_dev()), torch.ones(C, device=_dev()),
                        0.1, 1e-5)
    _, mask = K.bn_apply(z, vec[2], vec[3], relu=True, want_mask=True)
    dy = _rand_act((N, Kc, P, Q), 33)
    w = (torch.randn(C, R, R, Kc, generator=g) * (R * R * Kc) ** -0.5).to(_dev()).to(torch.bfloat16)
    addend = _rand_act((N, C, H, W), 34) if with_addend else None
    amask = torch.randint(0, 256, (N * H * W, C // 8), generator=g, dtype=torch.uint8).to(_dev()) if with_addend else None
    assert K.dgrad_bn_sums_ok(dy, w, stride, R, R)

    dx_ref = K.conv2d_dgrad(dy, w, (H, W), stride, pad, addend=addend, addend_mask=amask)
    ref = K.bn_backward(dx_ref, mask, z, gamma, vec[0], vec[1], relu=True)
    slab = K.new_slab(C, _dev())
    dx = K.conv2d_dgrad(dy, w, (H, W), stride, pad, addend=addend, addend_mask=amask, bn_sums=(z, mask, slab))
    assert torch.equal(dx, dx_ref)
    got = K.bn_backward(dx, mask, z, gamma, vec[0], vec[1], relu=True, slab=slab, presummed=True)
    assert float(slab[:2 * K._lib.bn_slots()].abs().max()) == 0.0      # self-cleaning: ready for the next step
    scale = max(1.0, float(ref[2].abs().max()), float(ref[3].abs().max()))
    _close(got[2], ref[2], rtol=2e-3, atol=2e-3 * scale, what="dgamma from epilogue sums")
    _close(got[3], ref[3], rtol=2e-3, atol=2e-3 * scale, what="dbeta from epilogue sums")
    _close(got[0], ref[0], rtol=2e-2, atol=2e-2, what="dx from epilogue sums")


# ----------------------------------------------------------------------------------------------- fused small MLP
@pytest.mark.gpu
@pytest.mark.parametrize("B,dims,relu_last", [(32, [48, 38, 38, 5], False), (300, [48, 38, 38, 38, 5], False),
                                              (129, [7, 64, 1], True), (1000, [64, 64, 64, 64, 10], False)])
def test_fused_mlp_matches_torch_fp32(B, dims, relu_last):
    """csrc/mlp.hip (whole Linear+ReLU stack in one forward and one backward launch) vs torch.nn in fp32."""
    from distributed_deep_learning_amd.nn.functional import FusedMLP
    torch.manual_seed(4)
    layers = [torch.nn.Linear(a, b).cuda() for a, b in zip(dims[:-1], dims[1:])]
    x = torch.randn(B, dims[0], device="cuda", requires_grad=True)
    h = x
    for i, l in enumerate(layers):
        h = l(h)
        if i + 1 < len(layers) or relu_last:
            h = torch.relu(h)
    w = torch.randn_like(h)
    (h * w).sum().backward()
    want = [x.grad.clone()] + [p.grad.clone() for l in layers for p in (l.weight, l.bias)]
    x.grad = None
    for l in layers:
        l.weight.grad = l.bias.grad = None
    params = [p for l in layers for p in (l.weight, l.bias)]
    out = FusedMLP.apply(x, relu_last, *params)
    assert torch.allclose(out, h.detach(), rtol=1e-4, atol=1e-5)
    (out * w).sum().backward()
    got = [x.grad] + [p.grad for p in params]
    for g, r in zip(got, want):
        assert g is not None and torch.allclose(g, r, rtol=2e-4, atol=2e-4 * float(r.abs().max() + 1e-6)), (g - r).abs().max()


@pytest.mark.gpu
def test_fused_mlp_accumulates_into_the_arena():
    from distributed_deep_learning_amd import models
    from distributed_deep_learning_amd.parallel import ParamArena
    torch.manual_seed(0)
    m = models.build("mlp").cuda()
    ref = models.build("mlp")
    ref.load_state_dict(m.state_dict())
    arena = ParamArena(m.named_parameters())
    x = torch.randn(200, 48)
    y = torch.randint(0, 5, (200,))
    torch.nn.functional.cross_entropy(ref(x), y).backward()
    out = m(x.cuda())
    assert "FusedMLP" in type(out.grad_fn).__name__
    torch.nn.functional.cross_entropy(out, y.cuda()).backward()
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert arena.grad_is_arena_view(p), n
        assert torch.allclose(p.grad.cpu(), q.grad, rtol=1e-3, atol=1e-5), n
