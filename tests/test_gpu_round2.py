"""Round-2 GPU checks: the kernel variants the headline benchmark actually runs (BK=32 ring tiles, >256-tile grids,
million-row filter gradients, the >4 GiB register-staged fallback), BatchNorm statistics under |mean| >> std, the
captured training step followed by an eager one, the default (native RCCL) reducer transport."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

from test_gpu_kernels import CL, _close, _dev, _rand_act

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# tiles = ceil(N*P*Q / 256) * ceil(K / 128) > 256 selects the BK=32 ring kernel (two resident blocks per CU): the variant
# that carries 15 of the 38.8 ms of the ResNet-50 step (profiles/r1_resnet50_b512_step_overlapped_kernel_summary_final.txt)
BENCH_SCALE_CASES = [
    # N, C, H, W, K, R, stride, pad
    (32, 64, 56, 56, 256, 1, 1, 0),      # layer1 conv3: 392 x 2 tiles, <128,32>
    (32, 256, 56, 56, 64, 1, 1, 0),      # layer1 conv1 of later blocks: 392 tiles, <64,32>
    (128, 128, 28, 28, 128, 3, 1, 1),    # layer2 conv2: 392 tiles, 36 K-steps of 32
    (16, 16, 115, 115, 64, 4, 1, 0),     # the space-to-depth stem at N=16: 784 tiles, <64,32>
    (128, 256, 56, 56, 128, 1, 2, 0),    # strided 1x1 (dgrad through parity classes): 392 tiles
]


@pytest.mark.parametrize("case", BENCH_SCALE_CASES)
def test_conv_kernels_at_benchmark_grid_sizes(case):
    from distributed_deep_learning_amd.ops import kernels as K

    N, C, H, W, Kc, R, stride, pad = case
    x = _rand_act((N, C, H, W), 1)
    w = (torch.randn((Kc, C, R, R), generator=torch.Generator().manual_seed(2)) / (C * R * R) ** 0.5)
    w = w.to(_dev()).contiguous(memory_format=CL)
    w_krsc, w_crsk = K.filter_repack(w)
    wq = w_krsc.permute(0, 3, 1, 2).float()
    P, Q = K.conv_out_size(H, R, stride, pad), K.conv_out_size(W, R, stride, pad)
    assert -(-N * P * Q // 256) * -(-Kc // 128) > 256, "case must select the BK=32 ring kernel"

    y, stats = K.conv2d_fwd(x, w_krsc, stride, pad, want_stats=True)
    y_ref = F.conv2d(x.float(), wq, None, stride, pad)
    _close(y, y_ref, what="conv fwd (BK=32)")
    s = K.slab_sums(stats)
    yf = y.float()
    _close(s[0], yf.sum((0, 2, 3)), rtol=2e-3, atol=1e-1 * max(1.0, yf.abs().max().item()), what="fused sum")
    _close(s[1], (yf * yf).sum((0, 2, 3)), rtol=2e-3, atol=1e-1, what="fused sumsq")
    dy = _rand_act(tuple(y.shape), 3)
    dx = K.conv2d_dgrad(dy, w_crsk, (H, W), stride, pad)
    dx_ref = torch.nn.grad.conv2d_input((N, C, H, W), wq, dy.float(), stride, pad)
    _close(dx, dx_ref, what="conv dgrad (BK=32)")
    dw = K.conv2d_wgrad(dy, x, R, R, stride, pad)
    dw_ref = torch.nn.grad.conv2d_weight(x.float(), (Kc, C, R, R), dy.float(), stride, pad)
    _close(dw, dw_ref, rtol=1e-2, atol=1e-2 * dw_ref.abs().max().item(), what="conv wgrad")


@pytest.mark.parametrize("env", [{"DDL_RING_BK": "32"}, {"DDL_RING_BK": "64"}, {"DDL_CONV_RING": "0"}],
                         ids=["ring-bk32", "ring-bk64", "register-staged"])
def test_every_conv_case_under_each_forced_kernel_variant(env):
    """The selection knobs are read once per process, so each variant gets a child process that runs ALL of CONV_CASES:
    BK=32 (and its 3*((row>>3)&1) swizzle) on small grids, BK=64 on every grid, and the register-staged conv_igemm
    kernel that large (> 4 GiB) operands fall back to."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_conv_cases_child.py")], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "0 failed" in r.stdout


def test_wgrad_over_a_million_rows_matches_fp32():
    """Split-M filter gradient with M = N*P*Q = 1.2 M rows (ResNet-50 layer1 at batch 384): fp32 atomics over thousands of
    partial tiles must still agree with an fp32 GEMM."""
    from distributed_deep_learning_amd.ops import kernels as K
    N, C, H, W, Kc = 384, 64, 56, 56, 64
    x = _rand_act((N, C, H, W), 5)
    dy = _rand_act((N, Kc, H, W), 6)
    dw = K.conv2d_wgrad(dy, x, 1, 1, 1, 0)
    xm = x.permute(0, 2, 3, 1).reshape(-1, C).float()
    dym = dy.permute(0, 2, 3, 1).reshape(-1, Kc).float()
    assert xm.shape[0] > 1_000_000
    ref = (dym.double().t() @ xm.double()).float().view(Kc, C, 1, 1)
    _close(dw, ref, rtol=2e-3, atol=2e-3 * ref.abs().max().item(), what="wgrad, 1.2M rows")
    # 3x3 at the same scale, accumulated INTO an existing gradient (the arena path)
    w3 = torch.full((Kc, C, 3, 3), 0.5, device=_dev()).contiguous(memory_format=CL)
    out = K.conv2d_wgrad(dy, x, 3, 3, 1, 1, out=w3)
    ref3 = torch.nn.grad.conv2d_weight(x[:64].float(), (Kc, C, 3, 3), dy[:64].float(), 1, 1)
    for i in range(64, N, 64):
        ref3 += torch.nn.grad.conv2d_weight(x[i:i + 64].float(), (Kc, C, 3, 3), dy[i:i + 64].float(), 1, 1)
    _close(out, ref3 + 0.5, rtol=5e-3, atol=5e-3 * ref3.abs().max().item(), what="wgrad 3x3 accumulate, 1.2M rows")


@pytest.mark.parametrize("case", [(3, 256, 15, 15, 384, 3, 1, 1), (5, 512, 9, 11, 128, 1, 1, 0), (2, 256, 20, 20, 256, 3, 2, 1)],
                         ids=["3x3-odd-rows", "1x1-plain", "3x3-stride2"])
def test_wide_filter_gradient_tiles_accumulate_into_an_existing_gradient(case):
    """K % 128 == 0 and S*C % 256 == 0 select the 128 x 256 tile (one block per CU, three ring stages, fp32 tile written
    back in two passes): row counts that are no multiple of the 32-pixel stage, several K tiles, and accumulation INTO a
    gradient that already holds values (the arena path)."""
    from distributed_deep_learning_amd.ops import kernels as K
    N, C, H, W, Kc, R, stride, pad = case
    assert Kc % 128 == 0 and (R * C) % 256 == 0
    P, Q = K.conv_out_size(H, R, stride, pad), K.conv_out_size(W, R, stride, pad)
    x = _rand_act((N, C, H, W), 11)
    dy = _rand_act((N, Kc, P, Q), 12)
    ref = torch.nn.grad.conv2d_weight(x.float(), (Kc, C, R, R), dy.float(), stride, pad)
    fresh = K.conv2d_wgrad(dy, x, R, R, stride, pad)
    _close(fresh, ref, rtol=1e-2, atol=1e-2 * ref.abs().max().item(), what="wide wgrad")
    held = torch.full((Kc, C, R, R), -0.25, device=_dev()).contiguous(memory_format=CL)
    out = K.conv2d_wgrad(dy, x, R, R, stride, pad, out=held)
    _close(out, ref - 0.25, rtol=1e-2, atol=1e-2 * ref.abs().max().item(), what="wide wgrad, accumulated")


def test_conv_with_an_operand_beyond_4_gib_takes_the_register_staged_kernel():
    """in_bytes >= 0xFFFF0000: 32-bit buffer offsets of the LDS-DMA ring kernel do not reach, the launcher must fall back
    to conv_igemm (64-bit addressing) and the result must still be right at the far end of the tensor."""
    from distributed_deep_learning_amd.ops import kernels as K
    N, C, H, W, Kc = 66, 64, 720, 720, 8
    assert N * C * H * W * 2 >= 0xFFFF0000
    g = torch.Generator(device=_dev()).manual_seed(3)
    x = torch.randn((N, H, W, C), generator=g, device=_dev(), dtype=torch.bfloat16).permute(0, 3, 1, 2)
    w = (torch.randn((Kc, C, 1, 1), generator=torch.Generator().manual_seed(2)) / C ** 0.5).to(_dev()).contiguous(memory_format=CL)
    w_krsc, w_crsk = K.filter_repack(w)
    y, _ = K.conv2d_fwd(x, w_krsc, 1, 0)
    wm = w_krsc.reshape(Kc, C).float()
    for n in (0, N // 2, N - 1):                         # first, middle and LAST image (past the 4 GiB mark)
        ref = (x[n].permute(1, 2, 0).reshape(-1, C).float() @ wm.t()).view(H, W, Kc).permute(2, 0, 1)
        _close(y[n], ref, what=f"conv fwd, image {n} of a 4.4 GB input")
    del y
    dy = torch.randn((N, H, W, Kc), generator=g, device=_dev(), dtype=torch.bfloat16).permute(0, 3, 1, 2)
    dx = K.conv2d_dgrad(dy, w_crsk, (H, W), 1, 0)          # OUTPUT beyond 4 GiB
    for n in (0, N - 1):
        ref = (dy[n].permute(1, 2, 0).reshape(-1, Kc).float() @ wm).view(H, W, C).permute(2, 0, 1)
        _close(dx[n], ref, what=f"conv dgrad, image {n}")


@pytest.mark.parametrize("mean,std", [(50.0, 0.1), (-300.0, 2.0), (0.5, 2.0)])
@pytest.mark.parametrize("fused", [False, True], ids=["bn_stats", "conv_epilogue"])
def test_bn_statistics_survive_a_mean_far_from_zero(mean, std, fused):
    """var = E[x^2] - E[x]^2 in fp32 cancels when |mean| >> std.  Reference: F.batch_norm on the same bf16-rounded data in
    fp32 (two-pass).  `fused`: statistics from the conv epilogue (identity 1x1 filter) instead of the stand-alone pass."""
    from distributed_deep_learning_amd.ops import kernels as K
    N, C, H, W = 64, 64, 28, 28
    g = torch.Generator().manual_seed(7)
    x = (torch.randn((N, C, H, W), generator=g) * std + mean).to(torch.bfloat16).to(_dev()).contiguous(memory_format=CL)
    gamma = (torch.rand(C, generator=g) + 0.5).to(_dev())
    beta = (torch.randn(C, generator=g) * 0.1).to(_dev())
    rm, rv = torch.zeros(C, device=_dev()), torch.ones(C, device=_dev())
    if fused:
        eye = torch.eye(C, device=_dev()).view(C, C, 1, 1).contiguous(memory_format=CL)
        wk, _ = K.filter_repack(eye)
        y, slab = K.conv2d_fwd(x, wk, 1, 0, want_stats=True)
        assert torch.equal(y, x)
    else:
        slab = K.bn_stats(x)
    vec = K.bn_finalize(slab, N * H * W, gamma, beta, rm, rv, 0.1, 1e-5)
    xf = x.float()
    mu_ref = xf.mean((0, 2, 3))
    var_ref = xf.var((0, 2, 3), unbiased=False)
    _close(vec[0], mu_ref, rtol=1e-5, atol=1e-5 * abs(mean) + 1e-6, what="batch mean")
    _close(vec[1], torch.rsqrt(var_ref + 1e-5), rtol=5e-3, atol=0.0, what="batch invstd")
    out = K.bn_apply(x, vec[2], vec[3], relu=False)
    ref = F.batch_norm(xf, None, None, gamma, beta, True, 0.1, 1e-5)
    _close(out, ref, rtol=2e-2, atol=3e-2, what="normalised output")


def test_eager_step_right_after_graph_capture_uses_live_weights():
    """GraphedTrainStep must invalidate the derived-weight caches it filled during capture (they live in the graph's
    private pool and were never executed): an eager evaluation right after the capture has to match a model that never
    captured anything (ragged batch after capture trained on uninitialised packs)."""
    from distributed_deep_learning_amd import models
    from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss
    from distributed_deep_learning_amd.optim import FusedSGD
    from distributed_deep_learning_amd.parallel import ParamArena
    from distributed_deep_learning_amd.utils import GraphedTrainStep

    def build():
        torch.manual_seed(11)
        m = models.build("lenet5").to(_dev())       # conv1 / conv2 use pad_out packs that are NOT arena-backed
        arena = ParamArena(m.named_parameters())
        return m, FusedSGD(arena, lr=0.05, momentum=0.9), SoftmaxCrossEntropyLoss()

    g = torch.Generator().manual_seed(1)
    x = torch.randn((32, 1, 28, 28), generator=g).to(_dev())
    y = torch.randint(0, 10, (32,), generator=g).to(_dev())
    xr = torch.randn((9, 1, 28, 28), generator=g).to(_dev())        # a ragged batch: runs eagerly
    yr = torch.randint(0, 10, (9,), generator=g).to(_dev())

    m1, o1, c1 = build()
    step = GraphedTrainStep(m1, c1, o1, x, y, warmup=1)             # one eager warm-up step, then the capture
    loss_eager_after_capture = c1(m1(xr), yr).item()                # eager forward BEFORE any replay
    m2, o2, c2 = build()
    c2(m2(x), y).backward()
    o2.step()                                                       # the same single (warm-up) step, never captured
    loss_ref = c2(m2(xr), yr).item()
    assert abs(loss_eager_after_capture - loss_ref) < 2e-2 * max(1.0, abs(loss_ref)), (loss_eager_after_capture, loss_ref)
    # and replays keep training the same model
    l0 = step(x, y).item()
    for _ in range(10):
        l1 = step(x, y).item()
    assert l1 < l0


def test_default_reducer_transport_is_the_native_communicator():
    """World 1 with the collectives forced: comm=None picks the native RCCL communicator on the GPU, its start-up check
    against torch.distributed passes, ncclCommCount reports the world, barrier() returns."""
    import torch.distributed as td
    from distributed_deep_learning_amd import dist as ddist
    from distributed_deep_learning_amd import models
    from distributed_deep_learning_amd.parallel import DistributedDataParallel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    ddist.init(ddist.RankInfo(0, 1, 0, 1), _dev(), force=True)
    try:
        m = models.build("resnet18_cifar").to(_dev())
        d = DistributedDataParallel(m, bucket_mb=8, force_collectives=True)
        assert d.ncomm is not None and d.comm_note is None
        assert d.comm_ranks() == 1 and d.ncomm.user_rank() == 0
        d.ncomm.barrier()
        # the communicator's stream was picked by measurement: it runs beside the compute stream AND the wgrad side stream
        from distributed_deep_learning_amd.ops import streams
        main = torch.cuda.current_stream(_dev())
        assert d.ncomm.stream.priority == 0
        assert d.ncomm.stream.cuda_stream != streams.side_stream(_dev()).cuda_stream
        assert streams._overlaps(main, d.ncomm.stream)                              # noqa: SLF001
        assert streams._overlaps(streams.side_stream(_dev()), d.ncomm.stream)       # noqa: SLF001
        d.close()
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("env", [{"DDL_CONV_PRING": "2", "DDL_PRING_MIN_TILES": "512"},
                                 {"DDL_CONV_PRING": "1", "DDL_PRING_MIN_TILES": "512"}, {"DDL_CONV_PRING": "0"}],
                         ids=["everything-it-can-run", "default-selection", "off"])
def test_persistent_ring_kernel_matches_fp32(env):
    """csrc/conv_pring.hip (persistent tiles, wave-private epilogue, stores left in flight behind a counted vmcnt, addend
    rows requested two rounds ahead, running fp64 statistics per block) against plain fp32 PyTorch: forward + statistics
    (1x1, 3x3, strided; channel tile changing inside a block's run), plain dgrad, dgrad with a masked and with a plain
    addend — 500+ tiles, ragged M and OC tails.  The same cases with the kernel switched off pin the comparison."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_pring_child.py")], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "0 failed" in r.stdout


@pytest.mark.parametrize("mode", ["2", "1"], ids=["everything-it-can-run", "default-selection"])
def test_big_tile_ping_pong_kernel_matches_fp32(mode):
    """csrc/conv_big.hip (256 x 256 tiles, two wave groups half a K-step apart, 128-byte LDS rows, persistent tiles,
    image-boundary split of a thin last tile round) against plain fp32 PyTorch on ragged shapes: forward + statistics
    (1x1, 3x3, strided), plain dgrad, dgrad with a masked / plain addend, and grids of 2 rounds + 18 tiles that the
    launcher cuts between this kernel and the ring kernels.  ``DDL_CONV_BIG`` is read once per process: child."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench", "big_probe.py"), "64", "parity", "child"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, DDL_CONV_BIG=mode))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "0 failed" in r.stdout and "FAIL" not in r.stdout


# --------------------------------------------------------------------------------------------- Linear on our GEMM
@pytest.mark.parametrize("B,fin,fout,relu,bias", [(128, 2048, 1000, False, True), (32, 25088, 4096, True, True),
                                                  (100, 400, 120, True, True), (9, 84, 10, False, True),
                                                  (64, 128, 6, False, False), (256, 4096, 4096, True, True)])
def test_linear_fwd_bwd_matches_fp32(B, fin, fout, relu, bias):
    """forward, input gradient, weight gradient and bias gradient of nn.Linear (all on ops.kernels.gemm_tn) against fp32
    F.linear on the bf16-rounded operands; covers VGG-16's 25088x4096, ragged batches and feature counts that are not
    multiples of 8."""
    from distributed_deep_learning_amd import nn as dnn
    g = torch.Generator().manual_seed(5)
    lin = dnn.Linear(fin, fout, bias=bias).to(_dev())
    with torch.no_grad():
        lin.weight.copy_(torch.randn((fout, fin), generator=g) / fin ** 0.5)
        if bias:
            lin.bias.copy_(torch.randn(fout, generator=g) * 0.1)
    x = torch.randn((B, fin), generator=g).to(torch.bfloat16).to(_dev()).requires_grad_(True)
    y = lin(x, relu=relu)
    assert y.shape == (B, fout) and y.dtype == torch.bfloat16
    wq = lin.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    bq = lin.bias.detach().clone().requires_grad_(True) if bias else None
    xr = x.detach().float().requires_grad_(True)
    ref = F.linear(xr, wq, bq)
    if relu:
        ref = F.relu(ref)
    _close(y, ref, what="linear fwd")
    dy = torch.randn((B, fout), generator=g).to(torch.bfloat16).to(_dev())
    y.backward(dy)
    # same gate as the kernel's own (bf16-rounded) output so rounding at 0 cannot flip elements between the two
    gate = (y.detach().float() > 0).float() if relu else torch.ones_like(ref)
    F.linear(xr, wq, bq).backward(dy.float() * gate)
    _close(x.grad, xr.grad, what="linear dx")
    _close(lin.weight.grad, wq.grad, rtol=1e-2, atol=1e-2 * wq.grad.abs().max().item(), what="linear dW")
    if bias:
        _close(lin.bias.grad, bq.grad, rtol=1e-2, atol=1e-2 * max(1.0, bq.grad.abs().max().item()), what="linear db")


def test_linear_gradients_go_straight_into_the_arena_and_dropout_gates_backward():
    from distributed_deep_learning_amd import nn as dnn
    from distributed_deep_learning_amd.parallel import ParamArena
    torch.manual_seed(3)
    lin = dnn.Linear(512, 256).to(_dev())
    arena = ParamArena(lin.named_parameters())
    assert lin.weight._ddl_shadow_t.shape == (512, 256)                        # the arena keeps the [in, out] twin
    assert torch.equal(lin.weight._ddl_shadow_t, lin.weight.detach().to(torch.bfloat16).t())
    x = torch.randn((64, 512), device=_dev()).to(torch.bfloat16).requires_grad_(True)
    lin.train()
    y = lin(x, relu=True, dropout=0.5)
    kept = (y.detach().float() > 0).float().mean().item()
    assert 0.15 < kept < 0.35, kept                                             # ~ half of the ~half that ReLU leaves
    y2 = lin(x, relu=True, dropout=0.5)
    assert not torch.equal(y, y2)                                               # the device-side counter advanced
    ref = F.relu(F.linear(x.detach().float(), lin.weight.detach().to(torch.bfloat16).float(), lin.bias.detach()))
    live = y.detach().float() > 0
    _close(y.detach().float()[live], 2.0 * ref[live], what="kept elements are scaled by 1/(1-p)")
    arena.zero_grad()
    y.backward(torch.ones_like(y))
    from distributed_deep_learning_amd.ops import streams
    streams.join()
    gq = live.float() * 2.0
    _close(lin.bias.grad, gq.sum(0), rtol=1e-2, atol=1e-1, what="db through the dropout gate")
    _close(lin.weight.grad, gq.t() @ x.detach().float(), rtol=1e-2, atol=2e-1, what="dW through the dropout gate")
    assert arena.grad_is_arena_view(lin.weight) and arena.grad_is_arena_view(lin.bias)
    lin.eval()
    _close(lin(x, relu=True, dropout=0.5), ref, what="eval mode: no dropout")


def test_linear_dropout_without_relu_keeps_the_gradient_of_negative_outputs():
    """Dropout behind a Linear with NO ReLU: kept elements may be negative, so the backward gate is ``y != 0`` (the
    dropped elements), not ``y > 0``.  The mask is read back from the forward output and the gradients are compared
    with fp32 autograd through that fixed mask."""
    from distributed_deep_learning_amd import nn as dnn
    torch.manual_seed(11)
    lin = dnn.Linear(384, 200).to(_dev())
    x = torch.randn((48, 384), device=_dev()).to(torch.bfloat16).requires_grad_(True)
    lin.train()
    y = lin(x, relu=False, dropout=0.25)
    yd = y.detach().float()
    mask = (yd != 0).float()
    assert 0.70 < mask.mean().item() < 0.80, mask.mean().item()
    assert (yd < 0).float().mean().item() > 0.25                                # negative outputs survive
    wq = lin.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    bq = lin.bias.detach().clone().requires_grad_(True)
    xr = x.detach().float().requires_grad_(True)
    ref = F.linear(xr, wq, bq) * mask / 0.75
    _close(y, ref, what="linear + dropout fwd")
    dy = torch.randn(y.shape, device=_dev()).to(torch.bfloat16)
    y.backward(dy)
    from distributed_deep_learning_amd.ops import streams
    streams.join()
    ref.backward(dy.float())
    _close(x.grad, xr.grad, what="dx through the dropout-only gate")
    _close(lin.weight.grad, wq.grad, rtol=1e-2, atol=1e-2 * wq.grad.abs().max().item(), what="dW through the dropout-only gate")
    _close(lin.bias.grad, bq.grad, rtol=1e-2, atol=1e-2 * max(1.0, bq.grad.abs().max().item()), what="db through the dropout-only gate")


def test_l1_loss_kernel_matches_torch():
    from distributed_deep_learning_amd import nn as dnn
    g = torch.Generator().manual_seed(9)
    for dtype in (torch.float32, torch.bfloat16):
        p = torch.randn((300, 5), generator=g).to(dtype).to(_dev()).requires_grad_(True)
        t = torch.randn((300, 5), generator=g).to(_dev())
        loss = dnn.L1Loss()(p, t)
        pr = p.detach().float().requires_grad_(True)
        ref = F.l1_loss(pr, t)
        assert abs(loss.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
        loss.backward()
        ref.backward()
        _close(p.grad, pr.grad, rtol=1e-2, atol=1e-6, what="l1 grad")


def test_side_stream_fork_join_survives_more_streams_than_hardware_queues():
    """The side stream is always normal priority (a high-priority one deadlocks a fork/join once the process holds more
    streams than hardware queues: bench/probes/prio_stream_probe.py, docs/STATUS.md) and the old switch is gone."""
    from distributed_deep_learning_amd.ops import kernels as K
    from distributed_deep_learning_amd.ops import streams
    os.environ["DDL_WGRAD_STREAM_PRIORITY"] = "-1"                  # must be ignored
    try:
        extra = [torch.cuda.Stream(device=_dev()) for _ in range(8)]
        x = _rand_act((4, 64, 16, 16), 1)
        for i in range(600):
            y = K.relu_fwd(x)
            with streams.fork(y):
                z = K.relu_fwd(y)
            with torch.cuda.stream(extra[i % 8]):
                K.relu_fwd(x)
            if i % 50 == 49:
                streams.join()
        streams.join()
        torch.cuda.synchronize()
        assert streams._stream(0).priority == 0                     # noqa: SLF001
        assert torch.equal(z.float(), F.relu(x.float()))
    finally:
        os.environ.pop("DDL_WGRAD_STREAM_PRIORITY", None)


@pytest.mark.parametrize("B", [3, 32, 700])
def test_conv1d_k1_relu_front_matches_torch(B):
    """The LSTM workload's front (Conv1d(10 -> 64, k=1) + ReLU on a [B, 10, 32] window) on the VALU kernel: forward and
    parameter gradients against fp32 torch."""
    from distributed_deep_learning_amd.nn import functional as DF
    g = torch.Generator().manual_seed(4)
    conv = torch.nn.Conv1d(10, 64, 1).to(_dev())
    x = torch.randn((B, 10, 32), generator=g).to(_dev())
    y = DF.Conv1dK1Relu.apply(x, conv.weight, conv.bias)
    ref = F.relu(conv(x))
    assert y.shape == (B, 64, 32) and y.dtype == torch.bfloat16
    _close(y, ref, what="conv1d k=1 fwd")
    dy = torch.randn((B, 64, 32), generator=g).to(torch.bfloat16).to(_dev())
    y.backward(dy)
    gw, gb = conv.weight.grad.clone(), conv.bias.grad.clone()
    conv.zero_grad()
    (conv(x) * (y.detach().float() > 0)).backward(dy.float())       # same gate as the kernel's own bf16 output
    _close(gw, conv.weight.grad, rtol=1e-2, atol=1e-2 * conv.weight.grad.abs().max().item(), what="conv1d dW")
    _close(gb, conv.bias.grad, rtol=1e-2, atol=1e-2 * conv.bias.grad.abs().max().item(), what="conv1d db")


def _dense_layers(c0, growth, nlayers, bn_size, dev, seed):
    from distributed_deep_learning_amd import nn as dnn

    torch.manual_seed(seed)
    layers, c = [], c0
    for _ in range(nlayers):
        n1 = dnn.BatchNorm2d(c)
        c1 = dnn.Conv2d(c, bn_size * growth, 1, 1, 0, bias=False)
        n2 = dnn.BatchNorm2d(bn_size * growth)
        c2 = dnn.Conv2d(bn_size * growth, growth, 3, 1, 1, bias=False)
        for bn in (n1, n2):
            torch.nn.init.uniform_(bn.weight, 0.5, 1.5)
            torch.nn.init.normal_(bn.bias, 0.0, 0.2)
        layers.append(tuple(m.to(dev) for m in (n1, c1, n2, c2)))
        c += growth
    return layers


def _close_l2(a, b, tol, what):
    """Gradients behind several bf16 ReLU gates: a pre-activation that rounds across zero flips a whole contribution, so
    single elements may be off while the tensor as a whole must agree (relative L2 error, plus a cap on outliers)."""
    a, b = a.float(), b.float()
    rel = ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    assert rel < tol, f"{what}: relative L2 error {rel:.4g} >= {tol}"
    if a.numel() < 4096:
        return
    bad = ((a - b).abs() > 0.1 * b.abs().max() + 0.1 * b.abs()).float().mean().item()
    assert bad < 5e-3, f"{what}: {bad:.4%} of the elements are off by more than 10 %"


def _dense_block_fp32(layers, x):
    """The reference's formulation: concatenate, BN-ReLU-conv1x1-BN-ReLU-conv3x3, in plain fp32 torch."""
    feats = [x]
    for n1, c1, n2, c2 in layers:
        h = torch.cat(feats, 1)
        h = F.relu(F.batch_norm(h, None, None, n1.weight, n1.bias, True, 0.0, n1.eps))
        h = F.conv2d(h, c1.weight.float())
        h = F.relu(F.batch_norm(h, None, None, n2.weight, n2.bias, True, 0.0, n2.eps))
        feats.append(F.conv2d(h, c2.weight.float(), padding=1))
    return torch.cat(feats, 1)


def _dense_block_layerwise(layers, x):
    """The same block through the per-layer modules (our kernels, concatenated features): identical precision."""
    from distributed_deep_learning_amd.nn.modules import conv_bn_act

    feats = [x]
    for n1, c1, n2, c2 in layers:
        h = n1(torch.cat(feats, 1) if len(feats) > 1 else feats[0], relu=True)
        feats.append(c2(conv_bn_act(c1, n2, h, relu=True)))
    return torch.cat(feats, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("N,c0,HW,growth,nlayers", [(4, 64, 16, 32, 2), (2, 64, 8, 32, 5), (3, 128, 12, 16, 3)])
def test_dense_block_on_one_feature_buffer_matches_fp32_concat_formulation(N, c0, HW, growth, nlayers):
    """Output against fp32 element-wise.  Gradients: a bf16 pre-activation that rounds across zero flips a whole
    term of a sum of random-sign terms, so against fp32 only ~sqrt(flip fraction) (5-14 % here, the layer-by-layer
    kernels show exactly the same) can be asked for; the sharp check is against the layer-by-layer path on the same
    kernels, which must agree to accumulation order."""
    from distributed_deep_learning_amd.nn.modules import dense_block

    dev = _dev()
    layers = _dense_layers(c0, growth, nlayers, 4, dev, 5)
    params = [p for l in layers for m in l for p in m.parameters()]
    x0 = _rand_act((N, c0, HW, HW), 21)
    gy = _rand_act((N, c0 + growth * nlayers, HW, HW), 22, 0.5)

    def run(fn, x, g):
        for p in params:
            p.grad = None
        x = x.clone().requires_grad_(True)
        y = fn(x)
        y.backward(g)
        return y.detach().float(), x.grad.float(), [p.grad.clone().float() for p in params]

    fused = run(lambda x: dense_block(layers, x), x0, gy)
    n1 = layers[-1][0]
    rmean, rvar = n1.running_mean.clone(), n1.running_var.clone()
    lw = run(lambda x: _dense_block_layerwise(layers, x), x0, gy)
    ref = run(lambda x: _dense_block_fp32(layers, x), x0.float(), gy.float())
    assert fused[0].shape == (N, c0 + growth * nlayers, HW, HW)
    _close(fused[0], ref[0], 3e-2, 3e-2, "dense block output")
    _close_l2(fused[1], lw[1], 1e-2, "input gradient vs layer-by-layer kernels")
    _close_l2(fused[1], ref[1], 1.5e-1, "input gradient vs fp32")
    for p, a, b, c in zip(params, fused[2], lw[2], ref[2]):
        _close_l2(a, b, 1e-2, f"parameter gradient {tuple(p.shape)} vs layer-by-layer kernels")
        _close_l2(a, c, 2e-1, f"parameter gradient {tuple(p.shape)} vs fp32")
    # running statistics of a late layer's first BatchNorm: taken from the shared totals table
    cat = ref[0][:, :n1.num_features]
    _close(rmean, 0.1 * cat.mean((0, 2, 3)), 2e-2, 2e-2, "norm1 running mean")
    _close(rvar, 0.9 + 0.1 * cat.var((0, 2, 3), unbiased=True), 2e-2, 2e-2, "norm1 running var")


@pytest.mark.gpu
def test_dense_block_eval_path_matches_layerwise_modules():
    from distributed_deep_learning_amd.nn.modules import conv_bn_act, dense_block

    dev = _dev()
    layers = _dense_layers(64, 32, 3, 4, dev, 6)
    for l in layers:
        for m in l:
            m.eval()
        for bn in (l[0], l[2]):
            bn.running_mean.normal_(0, 0.1)
            bn.running_var.uniform_(0.5, 1.5)
    x = _rand_act((2, 64, 8, 8), 23)
    with torch.no_grad():
        y = dense_block(layers, x)
        feats = [x]
        for n1, c1, n2, c2 in layers:
            h = n1(torch.cat(feats, 1), relu=True)
            feats.append(c2(conv_bn_act(c1, n2, h, relu=True)))
        ref = torch.cat(feats, 1)
    _close(y, ref, 1e-2, 1e-2, "dense block eval")


@pytest.mark.gpu
def test_channel_window_copy_roundtrip():
    from distributed_deep_learning_amd.ops import kernels as K

    buf = _rand_act((3, 96, 5, 7), 31)
    before = buf.clone()
    part = _rand_act((3, 32, 5, 7), 32)
    K.copy_channels(part, K.ChannelWindow(buf, 40, 32))
    assert torch.equal(buf[:, 40:72], part) and torch.equal(buf[:, :40], before[:, :40]) and torch.equal(buf[:, 72:], before[:, 72:])
    out = torch.empty_like(part)
    K.copy_channels(K.ChannelWindow(buf, 40, 32), out)
    assert torch.equal(out, part)
    with pytest.raises(ValueError):
        K.ChannelWindow(buf, 4, 32)


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,HW,Kc,R", [(32, 64, 56, 256, 1), (16, 128, 28, 128, 3), (8, 256, 14, 1024, 1), (4, 64, 9, 72, 3)])
def test_conv_with_fused_bn_finalize_equals_separate_finalize_every_time(N, C, HW, Kc, R):
    """The last-arriver finalize in the conv epilogue (csrc/conv_epilogue.h) against conv + stand-alone finalize, over
    repeated launches on the same self-cleaning slab: a stale slab read, a lost ticket or a counter that is not reset
    would show as a differing mean / invstd / running statistic in one of the rounds."""
    from distributed_deep_learning_amd.ops import kernels as K

    dev = _dev()
    g = torch.Generator().manual_seed(3)
    w = (torch.randn((Kc, C, R, R), generator=g) / (C * R * R) ** 0.5).to(dev).contiguous(memory_format=CL)
    wk, _ = K.filter_repack(w)
    gamma = (torch.rand(Kc, generator=g) + 0.5).to(dev)
    beta = torch.randn(Kc, generator=g).to(dev)
    slab_a, slab_b = K.new_slab(Kc, dev), K.new_slab(Kc, dev)
    rm_a, rv_a = torch.zeros(Kc, device=dev), torch.ones(Kc, device=dev)
    rm_b, rv_b = torch.zeros(Kc, device=dev), torch.ones(Kc, device=dev)
    pad = R // 2
    for it in range(6):
        x = _rand_act((N, C, HW, HW), 40 + it, 1.0 + it)
        y1, vec1 = K.conv2d_fwd_bn(x, wk, 1, pad, gamma, beta, rm_a, rv_a, 0.1, 1e-5, slab_a)
        y2, st = K.conv2d_fwd(x, wk, 1, pad, want_stats=True, slab=slab_b)
        vec2 = K.bn_finalize(st, N * HW * HW, gamma, beta, rm_b, rv_b, 0.1, 1e-5)
        assert torch.equal(y1, y2)
        torch.testing.assert_close(vec1, vec2, rtol=1e-5, atol=1e-6, msg=lambda m: f"round {it}: {m}")
        torch.testing.assert_close(rm_a, rm_b, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rv_a, rv_b, rtol=1e-5, atol=1e-6)
        assert not slab_a.any(), "slab and arrival counters must be left zero"
    ref = F.conv2d(x.float(), wk.float().permute(0, 3, 1, 2), padding=pad)
    _close(vec1[0], ref.mean((0, 2, 3)), 2e-2, 2e-2, "mean vs fp32")
    _close(vec1[1], (ref.var((0, 2, 3), unbiased=False) + 1e-5).rsqrt(), 2e-2, 2e-2, "invstd vs fp32")


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,HW,relu", [(32, 64, 56, True), (16, 128, 28, True), (16, 256, 14, False), (8, 512, 7, True),
                                         (8, 1024, 7, True), (4, 96, 9, True)])
def test_bn_backward_with_fused_finalize_matches_fp32_every_time(N, C, HW, relu):
    """bn_backward (reduce with last-arriver finalize for C <= 512, separate finalize above) over repeated launches on
    one self-cleaning slab, gradients accumulated into caller-owned tensors, against autograd in fp32."""
    from distributed_deep_learning_amd.ops import kernels as K

    dev = _dev()
    g = torch.Generator().manual_seed(9)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = (torch.randn(C, generator=g) * 0.1).to(dev)
    slab_f, slab_b = K.new_slab(C, dev), K.new_slab(C, dev)
    dg_acc, db_acc = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dg_ref, db_ref = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    for it in range(4):
        x = _rand_act((N, C, HW, HW), 60 + it, 1.5)
        dy = _rand_act((N, C, HW, HW), 70 + it)
        vec = K.bn_finalize(K.bn_stats(x, slab=slab_f), N * HW * HW, gamma, beta, None, None, 0.1, 1e-5)
        y, mask = K.bn_apply(x, vec[2], vec[3], relu=relu, want_mask=True)
        dx, _, _, _ = K.bn_backward(dy, mask, x, gamma, vec[0], vec[1], relu=relu, slab=slab_b, grad_out=(dg_acc, db_acc))
        nsl = 2 * K._lib.bn_slots()
        assert not slab_b[:nsl].any(), "the sums must be left zero"
        assert not slab_b[nsl:].reshape(-1)[3 * C:].any(), "the arrival counter must be left zero"
        xr = x.float().requires_grad_(True)
        gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        yr = F.batch_norm(xr, None, None, gr, br, True, 0.0, 1e-5)
        gate = (y.float() > 0).float() if relu else 1.0
        yr.backward(dy.float() * gate)
        dg_ref += gr.grad
        db_ref += br.grad
        _close(dx, xr.grad, 3e-2, 3e-2, f"dx round {it}")
        _close(dg_acc, dg_ref, 2e-2, 2e-2 * dg_ref.abs().max().item(), f"dgamma round {it}")
        _close(db_acc, db_ref, 2e-2, 2e-2 * db_ref.abs().max().item(), f"dbeta round {it}")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["2", "0"], ids=["always-fused", "never-fused"])
def test_fused_finalize_forced_on_and_off_in_a_child_process(mode):
    """DDL_FUSED_FINALIZE is read once per process and its default fuses only small tensors: force it on (2) for the
    large multi-XCD grids as well, and off (0), and run the finalize checks under each."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fused_finalize_child.py")], capture_output=True,
                       text=True, timeout=600, env=dict(os.environ, DDL_FUSED_FINALIZE=mode))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "0 failed" in r.stdout
