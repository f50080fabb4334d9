"""Raw (non-autograd) Python entry points of the gfx950 kernel library.

Conventions
-----------
* Activations are **bf16, logical NCHW, physically NHWC** (``torch.channels_last``).
* Filters are logical ``[K, C, R, S]`` with channels-last strides, i.e. physically ``[K][R][S][C]`` ("KRSC");
  the dgrad kernel wants the ``[C][R][S][K]`` repack ("CRSK") produced by :func:`filter_repack`.
* Every function launches on the current HIP stream and allocates through torch's caching allocator, so
  nothing here synchronises with the host.
"""
from __future__ import annotations

import torch

from . import _lib
from . import state as _state
from ._lib import check, ptr, stream_ptr

CL = torch.channels_last
BF16 = torch.bfloat16


def nhwc(x: torch.Tensor) -> torch.Tensor:
    """Return ``x`` as a channels-last-contiguous tensor (no copy when it already is)."""
    if x.dim() != 4:
        raise ValueError(f"expected a 4-D NCHW tensor, got {tuple(x.shape)}")
    return x if x.is_contiguous(memory_format=CL) else x.contiguous(memory_format=CL)


def _chk_act(x: torch.Tensor, name: str) -> None:
    if not x.is_cuda or x.dtype != BF16:
        raise TypeError(f"{name}: expected a bf16 GPU tensor, got {x.dtype} on {x.device}")
    _lib.check_device(x, name)              # launches go to the CURRENT device's stream (ops/_lib.py)


def conv_out_size(h: int, k: int, stride: int, pad: int) -> int:
    return (h + 2 * pad - k) // stride + 1


# ----------------------------------------------------------------------------------------------- conv
def filter_repack(w: torch.Tensor, want_copy: bool = True):
    """fp32/bf16 filter (logical KCRS, physically KRSC) -> (bf16 KRSC copy or None, bf16 CRSK repack)."""
    lib = _lib.require()
    K, Cc, R, S = w.shape
    wp = w.permute(0, 2, 3, 1)
    if not wp.is_contiguous():
        wp = wp.contiguous()
    is_f32 = w.dtype == torch.float32
    if not is_f32 and w.dtype != BF16:
        raise TypeError(f"filter_repack: unsupported dtype {w.dtype}")
    copy = torch.empty((K, R, S, Cc), dtype=BF16, device=w.device) if (want_copy and is_f32) else None
    wT = torch.empty((Cc, R, S, K), dtype=BF16, device=w.device)
    check(lib.ddl_filter_repack(ptr(wp), int(is_f32), ptr(copy), ptr(wT), K, R * S, Cc, stream_ptr()), "filter_repack")
    if copy is None and want_copy:
        copy = wp
    return copy, wT


def new_slab(channels: int, device) -> torch.Tensor:
    """A zeroed reduction slab: ``[slots, 2, C]`` fp64 sums (``csrc/common.h``: the variance subtraction needs them)
    held in an fp32 tensor of ``2*slots + 2`` rows.  Slabs are self-cleaning (the finalize kernels re-zero them), so
    one can be kept per layer and reused every step without a memset."""
    # two extra fp32 rows: scratch for the three per-channel coefficient vectors of the backward pass (never folded)
    return torch.zeros((2 * _lib.bn_slots() + 2, 2, channels), dtype=torch.float32, device=device)


def _slab_scratch(slab: torch.Tensor, channels: int) -> int:
    """Device address of the slab's fp32 scratch rows (behind the fp64 sums)."""
    return slab.data_ptr() + _lib.bn_slots() * 2 * channels * 8


def slab_sums(slab: torch.Tensor) -> torch.Tensor:
    """Totals ``[2, C]`` (fp32) of a reduction slab that has not been folded yet (tests / debugging)."""
    n = _lib.bn_slots()
    return slab[:2 * n].reshape(-1).view(torch.float64).view(n, 2, -1).sum(0).float()


def conv2d_fwd(x, w_krsc, stride=1, pad=0, bias=None, relu=False, want_stats=False, slab=None):
    """x: bf16 NHWC activations; w_krsc: bf16 ``[K,R,S,C]`` contiguous.  Returns ``(y, stats_slab or None)``.
    ``slab``: optional zeroed workspace from :func:`new_slab` to accumulate the BN partial sums into."""
    lib = _lib.require()
    _chk_act(x, "conv2d_fwd")
    x = nhwc(x)
    N, Cc, H, W = x.shape
    K, R, S, C2 = w_krsc.shape
    if C2 != Cc:
        raise ValueError(f"conv2d_fwd: input has {Cc} channels, filter expects {C2}")
    if w_krsc.dtype != BF16 or not w_krsc.is_contiguous():
        raise TypeError("conv2d_fwd: filter must be a contiguous bf16 [K,R,S,C] tensor")
    P, Q = conv_out_size(H, R, stride, pad), conv_out_size(W, S, stride, pad)
    if P <= 0 or Q <= 0:
        raise ValueError("conv2d_fwd: empty output")
    y = torch.empty((N, K, P, Q), dtype=BF16, device=x.device, memory_format=CL)
    stats = (slab if slab is not None else new_slab(K, x.device)) if want_stats else None
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != K):
        raise TypeError("conv2d_fwd: bias must be fp32 [K]")
    check(lib.ddl_conv_igemm(0, ptr(x), ptr(w_krsc), ptr(y), ptr(bias), ptr(stats), None, None, 1, None, None, 0,
                             N, P, Q, H, W, Cc, K, R, S, stride, pad, int(relu), stream_ptr()), "conv2d_fwd")
    return y, stats


def conv2d_fwd_bn(x, w_krsc, stride, pad, gamma, beta, running_mean, running_var, momentum, eps, slab):
    """``conv2d_fwd(want_stats=True)`` + :func:`bn_finalize` as ONE launch: the last block of every output-channel
    tile folds the statistics slab and writes ``vec[4, K]`` = (mean, invstd, scale, shift), running statistics included
    (``csrc/conv_epilogue.h``, last-arriver hand-off; kernels without it get the finalize launched behind them by the
    library).  ``slab``: this layer's zeroed :func:`new_slab` workspace (left zero).  Returns ``(y, vec)``."""
    lib = _lib.require()
    _chk_act(x, "conv2d_fwd_bn")
    x = nhwc(x)
    N, Cc, H, W = x.shape
    K, R, S, C2 = w_krsc.shape
    if C2 != Cc:
        raise ValueError(f"conv2d_fwd_bn: input has {Cc} channels, filter expects {C2}")
    if w_krsc.dtype != BF16 or not w_krsc.is_contiguous():
        raise ValueError("conv2d_fwd_bn: filter must be a contiguous bf16 [K,R,S,C] tensor")
    P, Q = conv_out_size(H, R, stride, pad), conv_out_size(W, S, stride, pad)
    y = torch.empty((N, K, P, Q), dtype=BF16, device=x.device, memory_format=CL)
    vec = torch.empty((4, K), dtype=torch.float32, device=x.device)
    check(lib.ddl_conv_fwd_bn(ptr(x), ptr(w_krsc), ptr(y), 0, ptr(slab), N, P, Q, H, W, Cc, K, R, S, stride, pad, ptr(gamma),
                              ptr(beta), ptr(running_mean), ptr(running_var), float(momentum), float(eps), ptr(vec),
                              stream_ptr()), "conv2d_fwd_bn")
    return y, vec


# ------------------------------------------------------------------------------- channel windows (dense blocks)
class ChannelWindow:
    """Channels ``[c0, c0 + c)`` of a bf16 NHWC buffer ``[N, Ctot, H, W]`` (channels_last) — what a dense block's
    layers read and write instead of concatenated copies (``/root/reference/src/pytorch/CNN/model.py:43-47`` re-copies
    every earlier feature map for every layer).  Only an address and a row pitch: no tensor view is created."""
    __slots__ = ("buf", "c0", "c", "ld", "rows", "shape")

    def __init__(self, buf: torch.Tensor, c0: int, c: int):
        _chk_act(buf, "ChannelWindow")
        N, Ctot, H, W = buf.shape
        if not buf.is_contiguous(memory_format=CL):
            raise ValueError("ChannelWindow: the buffer must be channels_last contiguous")
        if c0 % 8 or c % 8 or Ctot % 8 or c0 < 0 or c <= 0 or c0 + c > Ctot:
            raise ValueError(f"ChannelWindow: window [{c0}, {c0 + c}) of {Ctot} channels must be 8-aligned and in range")
        self.buf, self.c0, self.c, self.ld, self.rows, self.shape = buf, c0, c, Ctot, N * H * W, (N, c, H, W)

    def ptr(self) -> int:
        return self.buf.data_ptr() + 2 * self.c0


def copy_channels(src, dst):
    """``dst <- src`` where each side is a dense bf16 NHWC tensor or a :class:`ChannelWindow` of the same shape."""
    lib = _lib.require()

    def side(t):
        if isinstance(t, ChannelWindow):
            return t.ptr(), t.ld, t.shape
        _chk_act(t, "copy_channels")
        if not t.is_contiguous(memory_format=CL):
            raise ValueError("copy_channels: dense operands must be channels_last contiguous")
        return t.data_ptr(), t.shape[1], tuple(t.shape)

    sp, sld, ss = side(src)
    dp, dld, ds = side(dst)
    if ss != ds:
        raise ValueError(f"copy_channels: shapes differ, {ss} vs {ds}")
    check(lib.ddl_copy_channels(sp, sld, dp, dld, ss[0] * ss[2] * ss[3], ss[1], stream_ptr()), "copy_channels")
    return dst


def conv2d_fwd_into(x, w_krsc, window: ChannelWindow, stride=1, pad=0, slab=None, want_stats=True):
    """Forward convolution whose result lands in ``window`` (a channel window of a wider buffer) straight from the
    kernel's epilogue; with ``want_stats`` the statistics of the new channels go to ``slab``.  Returns the slab."""
    lib = _lib.require()
    _chk_act(x, "conv2d_fwd_into")
    x = nhwc(x)
    N, Cc, H, W = x.shape
    K, R, S, C2 = w_krsc.shape
    P, Q = conv_out_size(H, R, stride, pad), conv_out_size(W, S, stride, pad)
    if C2 != Cc or w_krsc.dtype != BF16 or not w_krsc.is_contiguous():
        raise ValueError("conv2d_fwd_into: needs a contiguous bf16 [K,R,S,C] filter matching the input channels")
    if window.shape != (N, K, P, Q):
        raise ValueError(f"conv2d_fwd_into: window {window.shape} is not the conv output {(N, K, P, Q)}")
    stats = (slab if slab is not None else new_slab(K, x.device)) if want_stats else None
    check(lib.ddl_conv_fwd_ld(ptr(x), ptr(w_krsc), window.ptr(), window.ld, ptr(stats), N, P, Q, H, W, Cc, K, R, S, stride,
                              pad, stream_ptr()), "conv2d_fwd_into")
    return stats


def bn_stats_window(win: ChannelWindow, slab=None):
    lib = _lib.require()
    if slab is None:
        slab = new_slab(win.c, win.buf.device)
    check(lib.ddl_bn_stats_ld(win.ptr(), win.ld, ptr(slab), win.rows, win.c, stream_ptr()), "bn_stats_window")
    return slab


def new_totals(channels: int, device) -> torch.Tensor:
    """Per-block statistics table ``[2, Ctot]`` fp64 (sum, sum of squares per channel of a dense block's buffer)."""
    return torch.empty((2, channels), dtype=torch.float64, device=device)


def slab_fold_totals(slab, channels: int, totals, c0: int):
    """Fold a statistics slab of ``channels`` channels (conv epilogue / :func:`bn_stats_window`) into columns
    ``[c0, c0 + channels)`` of ``totals``; the slab is zero afterwards."""
    lib = _lib.require()
    check(lib.ddl_slab_fold_totals(ptr(slab), channels, ptr(totals), totals.shape[1], c0, stream_ptr()), "slab_fold_totals")


def bn_finalize_totals(totals, channels: int, count, gamma, beta, running_mean, running_var, momentum, eps):
    """:func:`bn_finalize` for the first ``channels`` channels of a block's totals table (which stays intact)."""
    lib = _lib.require()
    vec = torch.empty((4, channels), dtype=torch.float32, device=totals.device)
    v0, row = vec.data_ptr(), 4 * channels
    check(lib.ddl_bn_finalize_totals(ptr(totals), totals.shape[1], channels, float(count), ptr(gamma), ptr(beta),
                                     ptr(running_mean), ptr(running_var), float(momentum), float(eps), v0, v0 + row,
                                     v0 + 2 * row, v0 + 3 * row, stream_ptr()), "bn_finalize_totals")
    return vec


def bn_apply_window(win: ChannelWindow, scale, shift, relu=True, want_mask=True):
    """Dense ``y = act(win * scale + shift)`` plus (``want_mask``) the ReLU bit mask, reading the window in place."""
    lib = _lib.require()
    N, Cc, H, W = win.shape
    y = torch.empty((N, Cc, H, W), dtype=BF16, device=win.buf.device, memory_format=CL)
    mask = torch.empty((win.rows, Cc // 8), dtype=torch.uint8, device=win.buf.device) if (relu and want_mask) else None
    check(lib.ddl_bn_apply_ld(win.ptr(), win.ld, ptr(y), ptr(mask), ptr(scale), ptr(shift), win.rows, Cc, int(relu),
                              stream_ptr()), "bn_apply_window")
    return y, mask


def bn_backward_window(dy, mask, xwin: ChannelWindow, gamma, mean, invstd, dxwin: ChannelWindow, add_to=None,
                       relu=True, slab=None, grad_out=None):
    """BatchNorm backward whose input was ``xwin``; the input gradient goes to ``dxwin``.  ``add_to`` (a window of the
    same position and pitch, possibly ``dxwin`` itself): ``dxwin = add_to + gradient``.  ``grad_out = (dgamma, dbeta)``
    fp32 tensors to accumulate into, else fresh ones are returned."""
    lib = _lib.require()
    _chk_act(dy, "bn_backward_window")
    dy = nhwc(dy)
    Cc = xwin.c
    if tuple(dy.shape) != xwin.shape or dxwin.shape != xwin.shape:
        raise ValueError("bn_backward_window: dy / x window / dx window shapes differ")
    if add_to is not None and (add_to.shape != dxwin.shape or add_to.ld != dxwin.ld or add_to.c0 != dxwin.c0):
        raise ValueError("bn_backward_window: add_to must be the same window of a buffer with the same pitch")
    if relu and (mask is None or mask.numel() != xwin.rows * Cc // 8):
        raise ValueError("bn_backward_window: relu=True needs the mask of bn_apply_window")
    if grad_out is not None:
        dg, db = grad_out
        if not (dg.is_contiguous() and db.is_contiguous() and dg.numel() == Cc and db.numel() == Cc
                and dg.dtype == torch.float32 and db.dtype == torch.float32):
            raise ValueError("bn_backward_window: grad_out must be two contiguous fp32 [C] tensors")
    else:
        t = torch.empty((2, Cc), dtype=torch.float32, device=dy.device)
        dg, db = t[0], t[1]
    if slab is None:
        slab = new_slab(Cc, dy.device)
    coef = _slab_scratch(slab, Cc)
    check(lib.ddl_bn_backward_ld(ptr(dy), ptr(mask) if relu else None, xwin.ptr(), xwin.ld, ptr(gamma), ptr(mean),
                                 ptr(invstd), dxwin.ptr(), dxwin.ld, add_to.ptr() if add_to is not None else None,
                                 ptr(dg), ptr(db), ptr(slab),
                                 ptr(coef), xwin.rows, Cc, int(relu), int(grad_out is not None), stream_ptr()),
          "bn_backward_window")
    return dg, db


def dgrad_bn_sums_ok(dy, w_crsk, stride, R, S) -> bool:
    """Whether :func:`conv2d_dgrad` can take ``bn_sums`` for this problem (ring kernel: 32-bit operand offsets; every
    input pixel produced by a launch, which excludes strides larger than the filter)."""
    return dy.numel() * 2 < 0xFFFF0000 and w_crsk.numel() * 2 < 0xFFFF0000 and (stride == 1 or (R >= stride and S >= stride))


def conv2d_dgrad(dy, w_crsk, in_hw, stride=1, pad=0, addend=None, addend_mask=None, addend_step=1, bn_sums=None,
                 relu_gate=None):
    """dy: bf16 NHWC ``[N,K,P,Q]``; w_crsk: bf16 ``[C,R,S,K]``; returns dx ``[N,C,H,W]`` (bf16 NHWC).
    ``addend`` (bf16, shape of dx) is added in the epilogue: gradient accumulation of a second path for free.
    ``addend_mask`` (uint8 ``[N*H*W, C/8]``, the ReLU bit mask :func:`bn_apply` writes) zeroes the addend where its
    bit is clear: ``dx = dgrad + addend * mask`` — the residual-branch gradient ``dy * relu'(y)`` without ever
    materialising it.  ``addend_step`` s > 1: the addend is COMPACT ``[N, C, ceil(H/s), ceil(W/s)]`` and is added at
    pixels ``(h, w)`` with ``h % s == w % s == 0`` only — the input gradient of a strided 1x1 projection shortcut,
    which is zero everywhere else, without the zero-filled full-size tensor.
    ``bn_sums = (z, mask or None, slab)``: dx is the gradient of ``act(BN(z))`` (``z`` bf16, shape of dx; ``mask`` its
    ReLU bit mask); the epilogue accumulates that BatchNorm's backward sums ``sum(g), sum(g*z)``, ``g = dx*mask``, into
    the zeroed ``slab`` (:func:`new_slab`), so :func:`bn_backward` can run with ``presummed=True`` (no reduce pass).
    ``relu_gate = (a, slab)``: dx is the gradient of a ReLU OUTPUT ``a`` (bf16, shape of dx — the conv's own input in a
    conv+bias+ReLU chain).  The epilogue stores ``dx * (a > 0)`` — the producer needs no ReLU-backward pass — and leaves
    its per-channel sums (= the producer's bias gradient) in row 0 of ``slab`` (:func:`slab_fold`)."""
    lib = _lib.require()
    _chk_act(dy, "conv2d_dgrad")
    dy = nhwc(dy)
    N, K, P, Q = dy.shape
    Cc, R, S, K2 = w_crsk.shape
    if K2 != K:
        raise ValueError("conv2d_dgrad: filter/grad channel mismatch")
    H, W = in_hw
    if conv_out_size(H, R, stride, pad) != P or conv_out_size(W, S, stride, pad) != Q:
        raise ValueError("conv2d_dgrad: geometry mismatch")
    dx = torch.empty((N, Cc, H, W), dtype=BF16, device=dy.device, memory_format=CL)
    if addend is not None:
        _chk_act(addend, "conv2d_dgrad(addend)")
        addend = nhwc(addend)
        want = dx.shape if addend_step == 1 else (N, Cc, -(-H // addend_step), -(-W // addend_step))
        if tuple(addend.shape) != tuple(want):
            raise ValueError("conv2d_dgrad: addend shape mismatch")
        if addend_step > 1 and addend_mask is not None:
            raise ValueError("conv2d_dgrad: a compact addend cannot be masked")
        if stride > 1 and (R < stride or S < stride):
            raise ValueError("conv2d_dgrad: addend not supported when stride exceeds the filter size")
    if addend_mask is not None:
        if addend is None:
            raise ValueError("conv2d_dgrad: addend_mask needs an addend")
        if addend_mask.dtype != torch.uint8 or addend_mask.numel() != N * H * W * Cc // 8 or not addend_mask.is_contiguous():
            raise ValueError("conv2d_dgrad: addend_mask must be the contiguous uint8 [N*H*W, C/8] mask of bn_apply")
    bz = bm = bslab = None
    flags = 0
    if relu_gate is not None:
        if bn_sums is not None:
            raise ValueError("conv2d_dgrad: bn_sums and relu_gate are exclusive")
        bn_sums = (relu_gate[0], None, relu_gate[1])
        flags = 3
    if bn_sums is not None:
        bz, bm, bslab = bn_sums
        bz = nhwc(bz)
        if tuple(bz.shape) != tuple(dx.shape) or bz.dtype != BF16:
            raise ValueError("conv2d_dgrad: bn_sums z must be bf16 with the shape of dx")
        if bm is not None and (bm.dtype != torch.uint8 or bm.numel() != N * H * W * Cc // 8 or not bm.is_contiguous()):
            raise ValueError("conv2d_dgrad: bn_sums mask must be the contiguous uint8 [N*H*W, C/8] mask of bn_apply")
        if not dgrad_bn_sums_ok(dy, w_crsk, stride, R, S):
            raise ValueError("conv2d_dgrad: bn_sums not available for this problem (see dgrad_bn_sums_ok)")
    check(lib.ddl_conv_igemm(1, ptr(dy), ptr(w_crsk), ptr(dx), None, ptr(bslab), ptr(addend), ptr(addend_mask),
                             int(addend_step), ptr(bz), ptr(bm), flags, N, H, W, P, Q, K, Cc, R, S, stride, pad, 0,
                             stream_ptr()), "conv2d_dgrad")
    return dx


_WGRAD_PLANS: dict = {}


def set_conv_c64_min_stages(n: int = 1024) -> None:
    """Test hook: the smallest stage count for which ``csrc/conv_c64.hip`` (3x3 over 64 -> 64 channels, forward and data
    gradient) is chosen instead of the ring kernels (default 1024 = four stages per block; 0: whenever the shape fits)."""
    _lib.require().ddl_conv_c64_min_stages(int(n))


def set_wgrad_halo_min_stages(n: int = 20) -> None:
    """Test hook: the shortest run of stages per block for which ``csrc/conv_wgrad.hip`` picks its halo kernel (default 20:
    small problems keep the ring kernels; 0: whenever the shape fits).  Clears the cached workspace plans."""
    _lib.require().ddl_wgrad_halo_min_stages(int(n))
    _WGRAD_PLANS.clear()


def _wgrad_workspace(lib, geom, device):
    """Scratch for the split-M partial tiles of one filter-gradient launch in REPRODUCIBLE mode (``ops.state``:
    ``set_deterministic`` / ``DDL_DETERMINISTIC=1`` / ``torch.use_deterministic_algorithms``): ``csrc/conv_wgrad.hip``
    then stores the partial tiles as slabs and adds them in a fixed order instead of meeting in float atomics, which
    makes every filter / Linear / LSTM weight gradient reproducible to the bit.  Sized by the launcher's own plan,
    cached per geometry; the buffer comes from the caching allocator on the CURRENT stream (the wgrad side stream), so
    consecutive launches reuse one block in stream order.  Returns ``(tensor or None, elements)``; ``(None, 0)`` selects
    the atomics (default: 0.6-1.3 % faster per ResNet-50 step, the slabs' extra bytes in an HBM-bound step)."""
    if not _state.deterministic():
        return None, 0
    need = _WGRAD_PLANS.get(geom)
    if need is None:
        need = int(lib.ddl_conv_wgrad_plan(*geom))
        if need < 0:
            raise ValueError(f"conv wgrad: unsupported problem {geom} (rc {need})")
        _WGRAD_PLANS[geom] = need
    if need == 0:
        return None, 0
    return torch.empty(need, dtype=torch.float32, device=device), need


def conv2d_wgrad(dy, x, R, S, stride=1, pad=0, out=None):
    """Returns dw as an fp32 tensor of logical shape ``[K,C,R,S]`` with channels-last (KRSC) strides.
    ``out``: optional fp32 ``[K,C,R,S]`` channels-last tensor to ACCUMULATE into (e.g. the arena gradient view)."""
    lib = _lib.require()
    _chk_act(dy, "conv2d_wgrad")
    _chk_act(x, "conv2d_wgrad")
    dy, x = nhwc(dy), nhwc(x)
    N, K, P, Q = dy.shape
    N2, Cc, H, W = x.shape
    if N2 != N or conv_out_size(H, R, stride, pad) != P or conv_out_size(W, S, stride, pad) != Q:
        raise ValueError("conv2d_wgrad: geometry mismatch")
    ws, ws_n = _wgrad_workspace(lib, (N, H, W, Cc, K, P, Q, R, S, stride, pad), x.device)
    if out is not None:
        if (out.dtype != torch.float32 or tuple(out.shape) != (K, Cc, R, S)
                or not out.permute(0, 2, 3, 1).is_contiguous()):
            raise ValueError("conv2d_wgrad: out must be fp32 [K,C,R,S] with channels-last (KRSC) strides")
        check(lib.ddl_conv_wgrad(ptr(dy), ptr(x), ptr(out), N, H, W, Cc, K, P, Q, R, S, stride, pad, ptr(ws), ws_n,
                                 stream_ptr()), "conv2d_wgrad")
        return out
    dw = torch.zeros((K, R, S, Cc), dtype=torch.float32, device=x.device)
    check(lib.ddl_conv_wgrad(ptr(dy), ptr(x), ptr(dw), N, H, W, Cc, K, P, Q, R, S, stride, pad, ptr(ws), ws_n,
                             stream_ptr()), "conv2d_wgrad")
    return dw.permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------------------------- batch norm
def bn_stats(x, slab=None):
    lib = _lib.require()
    _chk_act(x, "bn_stats")
    x = nhwc(x)
    N, Cc, H, W = x.shape
    if slab is None:
        slab = new_slab(Cc, x.device)
    check(lib.ddl_bn_stats(ptr(x), ptr(slab), N * H * W, Cc, stream_ptr()), "bn_stats")
    return slab


def bn_finalize(slab, count, gamma, beta, running_mean, running_var, momentum, eps):
    """slab ``[slots,2,C]`` -> ``vec[4,C]`` = (mean, invstd, scale, shift); updates running stats in place."""
    lib = _lib.require()
    Cc = slab.shape[-1]
    vec = torch.empty((4, Cc), dtype=torch.float32, device=slab.device)
    v0, row = vec.data_ptr(), 4 * Cc
    check(lib.ddl_bn_finalize(ptr(slab), _lib.bn_slots(), Cc, float(count), ptr(gamma), ptr(beta), ptr(running_mean),
                              ptr(running_var), float(momentum), float(eps), v0, v0 + row, v0 + 2 * row, v0 + 3 * row,
                              stream_ptr()), "bn_finalize")
    return vec


def vec_rows(vec):
    """Addresses of the four rows (mean, invstd, scale, shift) of a :func:`bn_finalize` result — accepted wherever a
    per-channel fp32 vector is expected, without creating tensor views on the hot path."""
    v0, row = vec.data_ptr(), 4 * vec.shape[1]
    return v0, v0 + row, v0 + 2 * row, v0 + 3 * row


def bn_apply(x, scale, shift, residual=None, relu=True, want_mask=False, res_scale=None, res_shift=None):
    """y = act(x*scale + shift [+ residual]).  With ``want_mask`` also returns the 1-bit-per-element ReLU mask
    (``uint8 [N*H*W, C/8]``) that :func:`bn_backward` consumes instead of re-reading ``y``.  ``res_scale`` /
    ``res_shift`` (fp32 ``[C]``): the residual enters as ``residual*res_scale + res_shift`` — a projection shortcut's
    BN applied on the fly to the raw shortcut-conv output."""
    lib = _lib.require()
    _chk_act(x, "bn_apply")
    x = nhwc(x)
    N, Cc, H, W = x.shape
    if residual is not None:
        _chk_act(residual, "bn_apply(residual)")
        residual = nhwc(residual)
        if residual.shape != x.shape:
            raise ValueError("bn_apply: residual shape mismatch")
    y = torch.empty_like(x, memory_format=CL)
    mask = torch.empty((N * H * W, Cc // 8), dtype=torch.uint8, device=x.device) if (want_mask and relu) else None
    check(lib.ddl_bn_apply(ptr(x), ptr(residual), ptr(y), ptr(mask), ptr(scale), ptr(shift), ptr(res_scale),
                           ptr(res_shift), N * H * W, Cc, int(relu), stream_ptr()), "bn_apply")
    return (y, mask) if want_mask else y


def bn_backward(dy, mask, x, gamma, mean, invstd, relu=True, want_dres=False, slab=None, grad_out=None,
                presummed=False):
    """Returns ``(dx, dres or None, dgamma, dbeta)``; ``mask`` is the bit mask from :func:`bn_apply` (``relu`` only).
    ``presummed``: ``slab`` already holds this BN's backward sums (left there by ``conv2d_dgrad(bn_sums=...)``)."""
    lib = _lib.require()
    _chk_act(dy, "bn_backward")
    dy, x = nhwc(dy), nhwc(x)
    N, Cc, H, W = x.shape
    if relu and (mask is None or mask.dtype != torch.uint8 or mask.numel() != N * H * W * Cc // 8):
        raise ValueError("bn_backward: relu=True needs the uint8 [N*H*W, C/8] mask from bn_apply(want_mask=True)")
    dx = torch.empty_like(x, memory_format=CL)
    dres = torch.empty_like(x, memory_format=CL) if want_dres else None
    if grad_out is not None:       # (dgamma, dbeta) fp32 [C] tensors to ACCUMULATE into (arena gradient views)
        dg, db = grad_out
        if not (dg.is_contiguous() and db.is_contiguous() and dg.numel() == Cc and db.numel() == Cc
                and dg.dtype == torch.float32 and db.dtype == torch.float32):
            raise ValueError("bn_backward: grad_out must be two contiguous fp32 [C] tensors")
        dgb = (dg, db)
    else:
        dgb = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    if slab is None:
        if presummed:
            raise ValueError("bn_backward: presummed needs the slab the sums were accumulated into")
        slab = new_slab(Cc, x.device)
    coef = _slab_scratch(slab, Cc)
    check(lib.ddl_bn_backward(ptr(dy), ptr(mask) if relu else None, ptr(x), ptr(gamma), ptr(mean), ptr(invstd), ptr(dx),
                              ptr(dres), ptr(dgb[0]), ptr(dgb[1]), ptr(slab), ptr(coef), N * H * W, Cc, int(relu),
                              int(grad_out is not None), int(presummed), stream_ptr()), "bn_backward")
    return dx, dres, dgb[0], dgb[1]


def slab_fold(slab, out, accumulate=False):
    """``out[c] (+)= sum over slots of slab[:, 0, c]``; the slab is zero afterwards (ready for the next use)."""
    lib = _lib.require()
    Cc = slab.shape[-1]
    if out.numel() != Cc or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("slab_fold: out must be a contiguous fp32 [C] tensor")
    check(lib.ddl_slab_fold(ptr(slab), Cc, ptr(out), int(accumulate), stream_ptr()), "slab_fold")
    return out


def bn_backward_pair(dy, mask, xa, xb, stat_a, stat_b, relu=True, slabs=None, grad_outs=None):
    """Backward of two BatchNorms fed by the same gradient ``dy`` (and ReLU ``mask``): a projection block's last BN
    (input ``xa``) and its shortcut BN (``xb``).  ``stat_x = (gamma, mean, invstd)``; ``slabs`` / ``grad_outs`` as in
    :func:`bn_backward`, one per BN.  ``dy`` and the mask are read once per pass for both.
    Returns ``(dxa, dxb, (dgamma_a, dbeta_a), (dgamma_b, dbeta_b))``."""
    lib = _lib.require()
    _chk_act(dy, "bn_backward_pair")
    dy, xa, xb = nhwc(dy), nhwc(xa), nhwc(xb)
    if xa.shape != xb.shape or dy.shape != xa.shape:
        raise ValueError("bn_backward_pair: dy / xa / xb shapes differ")
    N, Cc, H, W = xa.shape
    if relu and (mask is None or mask.dtype != torch.uint8 or mask.numel() != N * H * W * Cc // 8):
        raise ValueError("bn_backward_pair: relu=True needs the uint8 [N*H*W, C/8] mask from bn_apply(want_mask=True)")
    dxa = torch.empty_like(xa, memory_format=CL)
    dxb = torch.empty_like(xb, memory_format=CL)
    acc = grad_outs is not None
    if acc:
        for dg, db in grad_outs:
            if not (dg.is_contiguous() and db.is_contiguous() and dg.numel() == Cc and db.numel() == Cc
                    and dg.dtype == torch.float32 and db.dtype == torch.float32):
                raise ValueError("bn_backward_pair: grad_outs must be pairs of contiguous fp32 [C] tensors")
        ga, gb = grad_outs
    else:
        t = torch.empty((4, Cc), dtype=torch.float32, device=xa.device)
        ga, gb = (t[0], t[1]), (t[2], t[3])
    sa, sb = slabs if slabs is not None else (new_slab(Cc, xa.device), new_slab(Cc, xa.device))
    scratch = _lib.bn_slots() * 2 * Cc * 8
    check(lib.ddl_bn_backward_pair(ptr(dy), ptr(mask) if relu else None, ptr(xa), ptr(xb),
                                   ptr(stat_a[0]), ptr(stat_a[1]), ptr(stat_a[2]),
                                   ptr(stat_b[0]), ptr(stat_b[1]), ptr(stat_b[2]), ptr(dxa), ptr(dxb),
                                   ptr(ga[0]), ptr(ga[1]), ptr(gb[0]), ptr(gb[1]), ptr(sa), ptr(sb),
                                   sa.data_ptr() + scratch, sb.data_ptr() + scratch, N * H * W, Cc, int(relu), int(acc),
                                   stream_ptr()), "bn_backward_pair")
    return dxa, dxb, ga, gb


def bn_relu_maxpool_fwd(x, scale, shift, k, stride, pad, want_aux=False):
    """``maxpool(relu(x*scale + shift))`` straight from the conv output ``x``: returns ``(y_pool, idx)`` — the
    full-resolution activation is never written.  ``idx`` (uint8, shape of ``y_pool``) feeds
    :func:`bn_relu_maxpool_bwd`.  ``want_aux``: also returns ``(zmax, pmask)`` — the raw ``x`` at every argmax (bf16,
    shape of ``y_pool``) and one gate byte per 8 pooled channels — with which the backward takes its two sums over the
    pooled tensors instead of gathering 1-4 (gradient, argmax) pairs for every full-resolution pixel."""
    lib = _lib.require()
    _chk_act(x, "bn_relu_maxpool_fwd")
    x = nhwc(x)
    N, Cc, H, W = x.shape
    P, Q = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    if pad * 2 > k:
        raise ValueError("bn_relu_maxpool_fwd: padding larger than half the window")
    y = torch.empty((N, Cc, P, Q), dtype=BF16, device=x.device, memory_format=CL)
    idx = torch.empty((N, P, Q, Cc), dtype=torch.uint8, device=x.device)
    zmax = pmask = None
    if want_aux:
        zmax = torch.empty((N, P, Q, Cc), dtype=BF16, device=x.device)
        pmask = torch.empty((N * P * Q, Cc // 8), dtype=torch.uint8, device=x.device)
    check(lib.ddl_bn_relu_maxpool_fwd(ptr(x), ptr(scale), ptr(shift), ptr(y), ptr(idx), ptr(zmax), ptr(pmask), N, H, W, Cc,
                                      P, Q, k, stride, pad, stream_ptr()), "bn_relu_maxpool_fwd")
    return (y, idx, (zmax, pmask)) if want_aux else (y, idx)


def bn_relu_maxpool_bwd(dp, idx, x, gamma, mean, invstd, k, stride, pad, slab=None, grad_out=None, aux=None):
    """Backward of conv-output -> BN -> ReLU -> max-pool as one unit: ``dp`` is the gradient of the POOLED output.
    Returns ``(dx, dgamma, dbeta)`` with ``dx`` the gradient of the conv output ``x``.  ``aux``: the ``(zmax, pmask)``
    pair of :func:`bn_relu_maxpool_fwd` (pooled-domain reduction)."""
    lib = _lib.require()
    _chk_act(dp, "bn_relu_maxpool_bwd")
    dp, x = nhwc(dp), nhwc(x)
    N, Cc, H, W = x.shape
    _, _, P, Q = dp.shape
    if idx.dtype != torch.uint8 or idx.numel() != dp.numel() or not idx.is_contiguous():
        raise ValueError("bn_relu_maxpool_bwd: idx must be the uint8 argmax tensor of bn_relu_maxpool_fwd")
    dx = torch.empty_like(x, memory_format=CL)
    if grad_out is not None:
        dg, db = grad_out
        if not (dg.is_contiguous() and db.is_contiguous() and dg.numel() == Cc and db.numel() == Cc
                and dg.dtype == torch.float32 and db.dtype == torch.float32):
            raise ValueError("bn_relu_maxpool_bwd: grad_out must be two contiguous fp32 [C] tensors")
    else:
        both = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
        dg, db = both[0], both[1]
    if slab is None:
        slab = new_slab(Cc, x.device)
    coef = _slab_scratch(slab, Cc)
    zmax, pmask = aux if aux is not None else (None, None)
    if zmax is not None and (zmax.numel() != dp.numel() or zmax.dtype != BF16 or pmask.dtype != torch.uint8
                             or pmask.numel() * 8 != dp.numel() or not zmax.is_contiguous() or not pmask.is_contiguous()):
        raise ValueError("bn_relu_maxpool_bwd: aux must be the (zmax, pmask) pair of bn_relu_maxpool_fwd")
    check(lib.ddl_bn_relu_maxpool_bwd(ptr(dp), ptr(idx), ptr(x), ptr(zmax), ptr(pmask), ptr(gamma), ptr(mean), ptr(invstd), ptr(dx), ptr(dg),
                                      ptr(db), ptr(slab), coef, N, H, W, Cc, P, Q, k, stride, pad,
                                      int(grad_out is not None), stream_ptr()), "bn_relu_maxpool_bwd")
    return dx, dg, db


# ----------------------------------------------------------------------------------------------- pooling
def maxpool_fwd(x, k, stride, pad):
    lib = _lib.require()
    _chk_act(x, "maxpool_fwd")
    x = nhwc(x)
    N, Cc, H, W = x.shape
    P, Q = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    y = torch.empty((N, Cc, P, Q), dtype=BF16, device=x.device, memory_format=CL)
    idx = torch.empty((N, P, Q, Cc), dtype=torch.uint8, device=x.device)
    check(lib.ddl_maxpool_fwd(ptr(x), ptr(y), ptr(idx), N, H, W, Cc, P, Q, k, stride, pad, stream_ptr()), "maxpool_fwd")
    return y, idx


def maxpool_bwd(dy, idx, in_hw, k, stride, pad):
    lib = _lib.require()
    dy = nhwc(dy)
    N, Cc, P, Q = dy.shape
    H, W = in_hw
    dx = torch.empty((N, Cc, H, W), dtype=BF16, device=dy.device, memory_format=CL)
    check(lib.ddl_maxpool_bwd(ptr(dy), ptr(idx), ptr(dx), N, H, W, Cc, P, Q, k, stride, pad, stream_ptr()), "maxpool_bwd")
    return dx


def avgpool_fwd(x, k, stride):
    lib = _lib.require()
    _chk_act(x, "avgpool_fwd")
    x = nhwc(x)
    N, Cc, H, W = x.shape
    P, Q = conv_out_size(H, k, stride, 0), conv_out_size(W, k, stride, 0)
    y = torch.empty((N, Cc, P, Q), dtype=BF16, device=x.device, memory_format=CL)
    check(lib.ddl_avgpool_fwd(ptr(x), ptr(y), N, H, W, Cc, P, Q, k, stride, stream_ptr()), "avgpool_fwd")
    return y


def avgpool_bwd(dy, in_hw, k, stride):
    lib = _lib.require()
    dy = nhwc(dy)
    N, Cc, P, Q = dy.shape
    H, W = in_hw
    dx = torch.empty((N, Cc, H, W), dtype=BF16, device=dy.device, memory_format=CL)
    check(lib.ddl_avgpool_bwd(ptr(dy), ptr(dx), N, H, W, Cc, P, Q, k, stride, stream_ptr()), "avgpool_bwd")
    return dx


def gap_fwd(x):
    """Global average pool: ``[N,C,H,W]`` -> ``[N,C]`` (bf16)."""
    lib = _lib.require()
    _chk_act(x, "gap_fwd")
    x = nhwc(x)
    N, Cc, H, W = x.shape
    y = torch.empty((N, Cc), dtype=BF16, device=x.device)
    check(lib.ddl_gap_fwd(ptr(x), ptr(y), N, H * W, Cc, stream_ptr()), "gap_fwd")
    return y


def gap_bwd(dy, in_hw):
    lib = _lib.require()
    dy = dy.contiguous()
    N, Cc = dy.shape
    H, W = in_hw
    dx = torch.empty((N, Cc, H, W), dtype=BF16, device=dy.device, memory_format=CL)
    check(lib.ddl_gap_bwd(ptr(dy), ptr(dx), N, H * W, Cc, stream_ptr()), "gap_bwd")
    return dx


def filter_transpose_batched(src: torch.Tensor, dst: torch.Tensor, descs: torch.Tensor, ndesc: int, total_tiles: int):
    """One launch for every conv filter of an arena: bf16 KRSC shadow -> bf16 CRSK (see ParamArena)."""
    lib = _lib.require()
    check(lib.ddl_filter_transpose_batched(ptr(src), ptr(dst), ptr(descs), ndesc, total_tiles, stream_ptr()),
          "filter_transpose_batched")


# ----------------------------------------------------------------------------------------------- element-wise
def _flat_ok(*ts):
    n = ts[0].numel()
    for t in ts:
        if t.numel() != n or not (t.is_contiguous() or t.is_contiguous(memory_format=CL)):
            raise ValueError("element-wise kernels need dense tensors of equal size")
    if n % 8 != 0:
        raise ValueError("element-wise kernels need numel % 8 == 0")
    return n


def relu_fwd(x):
    lib = _lib.require()
    n = _flat_ok(x)
    y = torch.empty_like(x)
    check(lib.ddl_relu_fwd(ptr(x), ptr(y), n, stream_ptr()), "relu_fwd")
    return y


def relu_bwd(dy, y):
    lib = _lib.require()
    if dy.stride() != y.stride():
        dy = dy.contiguous(memory_format=CL) if y.is_contiguous(memory_format=CL) and y.dim() == 4 else dy.contiguous()
    n = _flat_ok(dy, y)
    dx = torch.empty_like(y)
    check(lib.ddl_relu_bwd(ptr(dy), ptr(y), ptr(dx), n, stream_ptr()), "relu_bwd")
    return dx


def add(a, b, relu=False):
    lib = _lib.require()
    if a.stride() != b.stride():
        b = b.contiguous(memory_format=CL) if a.is_contiguous(memory_format=CL) and a.dim() == 4 else b.contiguous()
    n = _flat_ok(a, b)
    out = torch.empty_like(a)
    check(lib.ddl_add(ptr(a), ptr(b), ptr(out), n, int(relu), stream_ptr()), "add")
    return out


def relu_bwd_bias_grad(dy, y, grad_out=None):
    """One pass: ``dx = dy * [y > 0]`` and the per-channel sums of ``dx`` (the bias gradient of a conv+bias+ReLU block).
    ``grad_out``: contiguous fp32 ``[C]`` tensor to ACCUMULATE into (arena gradient view); else a fresh tensor."""
    lib = _lib.require()
    _chk_act(dy, "relu_bwd_bias_grad")
    dy, y = nhwc(dy), nhwc(y)
    if y.shape != dy.shape:
        raise ValueError("relu_bwd_bias_grad: shape mismatch")
    N, Cc, H, W = dy.shape
    dx = torch.empty_like(dy, memory_format=CL)
    if grad_out is not None:
        if not (grad_out.is_contiguous() and grad_out.numel() == Cc and grad_out.dtype == torch.float32):
            raise ValueError("relu_bwd_bias_grad: grad_out must be a contiguous fp32 [C] tensor")
        db = grad_out
    else:
        db = torch.zeros(Cc, dtype=torch.float32, device=dy.device)     # zeroed by a fill kernel, kernel accumulates
    check(lib.ddl_relu_bwd_bias_grad(ptr(dy), ptr(y), ptr(dx), ptr(db), N * H * W, Cc, 1,
                                     stream_ptr()), "relu_bwd_bias_grad")
    return dx, db


def bias_grad(dy):
    """dy bf16 NHWC ``[N,K,P,Q]`` (or ``[M,K]``) -> fp32 ``[K]``."""
    lib = _lib.require()
    if dy.dim() == 4:
        dy = nhwc(dy)
        N, K, P, Q = dy.shape
        M = N * P * Q
    else:
        dy = dy.contiguous()
        M, K = dy.shape
    db = torch.empty((K,), dtype=torch.float32, device=dy.device)
    check(lib.ddl_bias_grad(ptr(dy), ptr(db), M, K, stream_ptr()), "bias_grad")
    return db


def cast_ranges(shadow: torch.Tensor, master: torch.Tensor, ranges: torch.Tensor, max_len: int) -> None:
    """``master[lo:hi] = float(shadow[lo:hi])`` for every ``[lo, hi)`` row of the int64 device table ``ranges`` - ONE
    launch (the PS / ZeRO-1 pull refreshes the fp32 masters of all foreign ranges with it)."""
    lib = _lib.require()
    if ranges.dtype != torch.int64 or ranges.dim() != 2 or ranges.shape[1] != 2 or not ranges.is_contiguous():
        raise ValueError("cast_ranges: ranges must be a contiguous int64 [n, 2] tensor")
    if shadow.dtype != BF16 or master.dtype != torch.float32 or shadow.numel() != master.numel():
        raise ValueError("cast_ranges: bf16 shadow and fp32 master of the same length")
    _lib.check_device(master, "cast_ranges")
    check(lib.ddl_cast_ranges(ptr(shadow), ptr(master), ptr(ranges), ranges.shape[0], int(max_len), stream_ptr()), "cast_ranges")


def pack_owned(master: torch.Tensor, idx: torch.Tensor, own: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """``out[i] = master[idx[i]] if own[i] else 0`` (fp32; ``idx`` int64, ``own`` bool / uint8) in one launch."""
    lib = _lib.require()
    n = idx.numel()
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=master.device)
    if idx.dtype != torch.int64 or own.numel() != n or own.element_size() != 1 or out.numel() != n or master.dtype != torch.float32:
        raise ValueError("pack_owned: bad operands")
    _lib.check_device(master, "pack_owned")
    check(lib.ddl_pack_owned(ptr(master), ptr(idx), ptr(own), ptr(out), n, stream_ptr()), "pack_owned")
    return out


def scatter_values(master: torch.Tensor, idx: torch.Tensor, vals: torch.Tensor) -> None:
    """``master[idx[i]] = vals[i]`` (fp32, unique indices) in one launch."""
    lib = _lib.require()
    if idx.dtype != torch.int64 or vals.numel() != idx.numel() or vals.dtype != torch.float32 or master.dtype != torch.float32:
        raise ValueError("scatter_values: bad operands")
    _lib.check_device(master, "scatter_values")
    check(lib.ddl_scatter_values(ptr(master), ptr(idx), ptr(vals), idx.numel(), stream_ptr()), "scatter_values")


def cast_f32_bf16(src: torch.Tensor, dst: torch.Tensor | None = None):
    lib = _lib.require()
    n = src.numel()
    if dst is None:
        dst = torch.empty(src.shape, dtype=BF16, device=src.device)
    check(lib.ddl_cast_f32_bf16(ptr(src), ptr(dst), n, stream_ptr()), "cast_f32_bf16")
    return dst


def cast_bf16_f32(src: torch.Tensor, dst: torch.Tensor | None = None):
    lib = _lib.require()
    n = src.numel()
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    check(lib.ddl_cast_bf16_f32(ptr(src), ptr(dst), n, stream_ptr()), "cast_bf16_f32")
    return dst


def nchw_to_nhwc_pad(x: torch.Tensor, cp: int):
    """fp32 NCHW-contiguous images -> bf16 NHWC with channels zero-padded to ``cp`` (logical ``[N,cp,H,W]``)."""
    lib = _lib.require()
    x = x.contiguous()
    N, Cc, H, W = x.shape
    out = torch.empty((N, cp, H, W), dtype=BF16, device=x.device, memory_format=CL)
    check(lib.ddl_nchw_to_nhwc_pad(ptr(x), ptr(out), N, Cc, H * W, cp, stream_ptr()), "nchw_to_nhwc_pad")
    return out


def nchw_to_s2d(x: torch.Tensor):
    """fp32 NCHW 3-channel images -> bf16 NHWC space-to-depth tensor ``[N, 16, H/2+3, W/2+3]`` (logical NCHW view)
    with the stem's padding materialised; see ``csrc/eltwise.hip::nchw_to_s2d_kernel`` for the channel order."""
    lib = _lib.require()
    x = x.contiguous()
    N, Cc, H, W = x.shape
    if Cc != 3 or x.dtype != torch.float32 or H % 2 or W % 2:
        raise ValueError("nchw_to_s2d: needs fp32 [N,3,H,W] with even H and W")
    out = torch.empty((N, 16, H // 2 + 3, W // 2 + 3), dtype=BF16, device=x.device, memory_format=CL)
    check(lib.ddl_nchw_to_s2d(ptr(x), ptr(out), N, H, W, stream_ptr()), "nchw_to_s2d")
    return out


def s2d_filter(w: torch.Tensor) -> torch.Tensor:
    """``[K,3,7,7]`` stem filter -> the equivalent ``[K,16,4,4]`` filter over :func:`nchw_to_s2d` input
    (tap ``r = 2a + p``; the 8th tap and channels 12..15 are zero).  Plain differentiable torch ops: the filter
    gradient flows back to the 7x7 parameter through autograd."""
    K_ = w.shape[0]
    w8 = torch.nn.functional.pad(w, (0, 1, 0, 1))                       # [K,3,8,8]
    w8 = w8.reshape(K_, 3, 4, 2, 4, 2).permute(0, 3, 5, 1, 2, 4)        # [K, p, q, c, a, b]
    w12 = w8.reshape(K_, 12, 4, 4)
    return torch.nn.functional.pad(w12, (0, 0, 0, 0, 0, 4))             # [K,16,4,4]


# ----------------------------------------------------------------------------------------------- linear (fc) layers
def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def gemm_tn(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """``out[Ka, Cb] += a[M, Ka]^T @ b[M, Cb]`` — bf16 operands with the REDUCTION index slowest, fp32 accumulation into
    ``out`` (contiguous; zero it for a plain product).  Runs on the split-M MFMA kernel of ``csrc/conv_wgrad.hip``
    (a filter gradient with 1x1 taps IS this GEMM), which spreads the reduction over the whole chip."""
    lib = _lib.require()
    _chk_act(a, "gemm_tn")
    _chk_act(b, "gemm_tn")
    M, Ka = a.shape
    M2, Cb = b.shape
    if M2 != M or Ka % 8 or Cb % 8 or not a.is_contiguous() or not b.is_contiguous():
        raise ValueError("gemm_tn: operands must be contiguous [M, Ka] / [M, Cb] with Ka, Cb multiples of 8")
    if out.dtype != torch.float32 or tuple(out.shape) != (Ka, Cb) or not out.is_contiguous():
        raise ValueError("gemm_tn: out must be a contiguous fp32 [Ka, Cb] tensor")
    ws, ws_n = _wgrad_workspace(lib, (M, 1, 1, Cb, Ka, 1, 1, 1, 1, 1, 0), a.device)
    check(lib.ddl_conv_wgrad(ptr(a), ptr(b), ptr(out), M, 1, 1, Cb, Ka, 1, 1, 1, 1, 1, 0, ptr(ws), ws_n, stream_ptr()),
          "gemm_tn")
    return out


def transpose_pad(x: torch.Tensor, rows_padded: int | None = None) -> torch.Tensor:
    """``[R, C]`` (bf16 or fp32, rows ``C``-contiguous) -> bf16 ``[C, Rp]`` with columns ``R..Rp-1`` zero."""
    lib = _lib.require()
    R, Cc = x.shape
    if x.stride(1) != 1 or x.dtype not in (BF16, torch.float32):
        raise ValueError("transpose_pad: need a bf16 / fp32 matrix with contiguous rows")
    _lib.check_device(x, "transpose_pad")
    Rp = _pad8(R) if rows_padded is None else rows_padded
    out = torch.empty((Cc, Rp), dtype=BF16, device=x.device)
    check(lib.ddl_transpose_pad(ptr(x), int(x.dtype == torch.float32), ptr(out), R, Cc, Rp, x.stride(0), stream_ptr()),
          "transpose_pad")
    return out


_RNG_STATE: dict = {}


_RNG_STREAM = 0          # which independent stream of masks this process draws (its data-parallel rank)


def _rng_seed() -> int:
    # splitmix-style mix of (torch seed, stream): ranks that share torch.manual_seed(s) still draw different masks
    z = (torch.initial_seed() + 0x9E3779B97F4A7C15 * (_RNG_STREAM + 1)) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return (z ^ (z >> 31)) & 0x7FFFFFFFFFFFFFFF


def rng_state(device) -> torch.Tensor:
    """Device-resident ``[seed, counter]`` of the dropout generator (seeded from ``torch.initial_seed()`` and the
    process's mask stream - :func:`seed_dropout` - on first use; the kernels advance the counter themselves, so captured
    graphs draw fresh masks on every replay)."""
    key = (device.type, device.index)
    st = _RNG_STATE.get(key)
    if st is None:
        st = _RNG_STATE[key] = torch.tensor([_rng_seed(), 0], dtype=torch.int64, device=device)
    return st


def seed_dropout(stream: int = 0) -> None:
    """(Re)seed the dropout generators of this process from ``torch.initial_seed()`` and ``stream`` (the data-parallel
    rank: replicas seeded alike by ``torch.manual_seed`` must not drop the same units of their different samples).
    Existing device states are rewritten IN PLACE (captured graphs hold their address); counters restart at 0."""
    global _RNG_STREAM
    _RNG_STREAM = int(stream)
    for st in _RNG_STATE.values():
        st.copy_(torch.tensor([_rng_seed(), 0], dtype=torch.int64))


def dropout_state_dict() -> dict:
    """``{"cuda:0": [seed, counter], ...}`` for checkpoints (plain ints)."""
    return {(k[0] if k[1] is None else f"{k[0]}:{k[1]}"): [int(v) for v in st.cpu().tolist()] for k, st in _RNG_STATE.items()}


def load_dropout_state_dict(sd: dict, counters_only: bool = True) -> None:
    """Restore the mask COUNTERS (and the seeds too with ``counters_only=False``): a resumed run continues the mask
    sequence where the checkpointed one stopped; the seed stays the one this rank was given."""
    for name, (seed, counter) in (sd or {}).items():
        dev = torch.device(name)
        if dev.type == "cuda" and (not torch.cuda.is_available() or dev.index >= torch.cuda.device_count()):
            continue
        st = rng_state(dev)
        cur = st.cpu().tolist()
        st.copy_(torch.tensor([cur[0] if counters_only else int(seed), int(counter)], dtype=torch.int64))


_ACCUMULATORS: dict = {}


def zero_accumulator(rows: int, cols: int, device) -> torch.Tensor:
    """A persistent fp32 ``[rows, cols]`` GEMM accumulator that is ALWAYS ZERO between uses: whoever takes it runs the
    split-M GEMM into it and then reads it out with ``linear_epilogue(..., clear=True)`` / ``cast_rows(..., clear=True)``,
    which zero it behind the read - no allocation and no memset launch per GEMM.  One per shape and device; all users are
    ordered on the device's current stream.  :func:`drop_accumulators` forgets them (after a failed launch)."""
    key = (rows, cols, device.type, device.index)
    acc = _ACCUMULATORS.get(key)
    if acc is None:
        if torch.cuda.is_current_stream_capturing():          # a graph's private pool must not own a persistent buffer
            return torch.zeros((rows, cols), dtype=torch.float32, device=device)
        acc = _ACCUMULATORS[key] = torch.zeros((rows, cols), dtype=torch.float32, device=device)
    return acc


def drop_accumulators() -> None:
    _ACCUMULATORS.clear()


def linear_epilogue(acc: torch.Tensor, bias, rows: int, relu: bool = False, drop_p: float = 0.0,
                    clear: bool = False) -> torch.Tensor:
    """fp32 accumulator ``[Mp, N]`` -> bf16 ``[rows, N]``: ``drop(relu(acc + bias))`` (``csrc/linear.hip``); ``clear``
    zeroes the whole accumulator behind the read."""
    lib = _lib.require()
    Mp, N = acc.shape
    if acc.dtype != torch.float32 or not acc.is_contiguous() or N % 8 or rows > Mp:
        raise ValueError("linear_epilogue: acc must be a contiguous fp32 [Mp, N] tensor with N % 8 == 0")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("linear_epilogue: bias must be a contiguous fp32 [N] tensor")
    _lib.check_device(acc, "linear_epilogue")
    y = torch.empty((rows, N), dtype=BF16, device=acc.device)
    st = rng_state(acc.device) if drop_p > 0.0 else None
    check(lib.ddl_linear_epilogue(ptr(acc), ptr(bias), ptr(y), rows, N, int(relu), float(drop_p), ptr(st),
                                  Mp if clear else 0, stream_ptr()), "linear_epilogue")
    return y


def linear_bwd_gate(dy: torch.Tensor, y, scale: float = 1.0, grad_out=None, want_bias: bool = True, want_g: bool = True,
                    gate_nonzero: bool = False):
    """``g = dy * [y > 0] * scale`` (``y`` given: ReLU, with or without dropout behind it), ``dy * [y != 0] * scale``
    (``gate_nonzero``: dropout WITHOUT ReLU - kept elements may be negative) or ``dy * scale``; the bias gradient
    ``sum_rows g`` goes to a fresh fp32 ``[C]`` tensor or is ACCUMULATED into ``grad_out``.  Returns ``(g, db)``."""
    lib = _lib.require()
    _chk_act(dy, "linear_bwd_gate")
    M, Cc = dy.shape
    if not dy.is_contiguous() or Cc % 8 or (y is not None and (y.shape != dy.shape or not y.is_contiguous())):
        raise ValueError("linear_bwd_gate: dy / y must be contiguous [M, C] with C % 8 == 0")
    g = torch.empty_like(dy) if want_g else None
    db = None
    if want_bias:
        if grad_out is not None:
            if grad_out.dtype != torch.float32 or grad_out.numel() != Cc or not grad_out.is_contiguous():
                raise ValueError("linear_bwd_gate: grad_out must be a contiguous fp32 [C] tensor")
            db = grad_out
        else:
            db = torch.empty(Cc, dtype=torch.float32, device=dy.device)
    check(lib.ddl_linear_bwd_gate(ptr(dy), ptr(y), ptr(g), ptr(db), M, Cc, float(scale), int(grad_out is not None),
                                  int(bool(gate_nonzero)), stream_ptr()), "linear_bwd_gate")
    return g, db


def cast_rows(acc: torch.Tensor, rows: int, cols: int, clear: bool = False) -> torch.Tensor:
    """fp32 ``[Mp, Np]`` accumulator -> bf16 ``[rows, cols]`` (drops the padding; ``cols % 8 == 0``); ``clear`` (needs
    ``cols == Np``) zeroes the whole accumulator behind the read."""
    lib = _lib.require()
    if acc.dtype != torch.float32 or not acc.is_contiguous() or cols % 8 or rows > acc.shape[0] or cols > acc.shape[1]:
        raise ValueError("cast_rows: bad shapes")
    if clear and cols != acc.shape[1]:
        raise ValueError("cast_rows: clear needs cols == the accumulator's width")
    out = torch.empty((rows, cols), dtype=BF16, device=acc.device)
    check(lib.ddl_cast_rows(ptr(acc), ptr(out), rows, cols, acc.shape[1], acc.shape[0] if clear else 0, stream_ptr()),
          "cast_rows")
    return out


def l1_loss(pred: torch.Tensor, target: torch.Tensor, want_grad: bool = True):
    """Mean absolute error: returns ``(loss fp32 [1], grad fp32 like pred or None)`` with ``grad = sign(p - t) / n``."""
    lib = _lib.require()
    pred = pred.contiguous()
    target = target.to(torch.float32).contiguous()
    if pred.dtype not in (BF16, torch.float32) or target.numel() != pred.numel():
        raise ValueError("l1_loss: pred must be bf16 / fp32 and match the target's size")
    _lib.check_device(pred, "l1_loss")
    out = torch.empty(1, dtype=torch.float32, device=pred.device)
    grad = torch.empty(pred.shape, dtype=torch.float32, device=pred.device) if want_grad else None
    check(lib.ddl_l1_loss(ptr(pred), int(pred.dtype == BF16), ptr(target), ptr(grad), ptr(out), pred.numel(), stream_ptr()),
          "l1_loss")
    return out, grad


def conv1d_k1_supported(C: int, O: int, L: int) -> bool:
    return 1 <= C <= 16 and L % 8 == 0 and O * (L // 8) <= 1024


def conv1d_k1_relu_fwd(x: torch.Tensor, w: torch.Tensor, bias) -> torch.Tensor:
    """``relu(conv1d(x, w[:, :, None], bias))`` for fp32 ``x [B, C, L]``, ``w [O, C]`` -> bf16 ``[B, O, L]``."""
    lib = _lib.require()
    B, Cc, L = x.shape
    O = w.shape[0]
    if x.dtype != torch.float32 or not x.is_contiguous() or w.dtype != torch.float32 or not w.is_contiguous() \
            or w.shape[1] != Cc or not conv1d_k1_supported(Cc, O, L):
        raise ValueError("conv1d_k1_relu_fwd: need contiguous fp32 x [B,C,L], w [O,C], C <= 16, L % 8 == 0")
    _lib.check_device(x, "conv1d_k1_relu_fwd")
    y = torch.empty((B, O, L), dtype=BF16, device=x.device)
    check(lib.ddl_conv1d_k1_relu_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), B, Cc, O, L, stream_ptr()), "conv1d_k1_relu_fwd")
    return y


def conv1d_k1_relu_bwd(dy: torch.Tensor, y: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, db) -> None:
    """Accumulate the weight (``[O, C]`` fp32) and bias gradients of :func:`conv1d_k1_relu_fwd` into ``dw`` / ``db``."""
    lib = _lib.require()
    B, O, L = y.shape
    Cc = x.shape[1]
    if dy.dtype != BF16 or not dy.is_contiguous() or dy.shape != y.shape or dw.dtype != torch.float32 \
            or not dw.is_contiguous() or dw.numel() != O * Cc:
        raise ValueError("conv1d_k1_relu_bwd: dy must be contiguous bf16 like y, dw contiguous fp32 [O, C]")
    check(lib.ddl_conv1d_k1_relu_bwd(ptr(dy), ptr(y), ptr(x), ptr(dw), ptr(db), B, Cc, O, L, stream_ptr()),
          "conv1d_k1_relu_bwd")


# ----------------------------------------------------------------------------------------------- fused small MLP
MLP_MAX_WIDTH, MLP_MAX_LAYERS, _MLP_LDS_FLOATS = 64, 8, (160 * 1024) // 4 - 3 * 64 * 33


def mlp_supported(dims) -> bool:
    """Layer widths ``dims = [in, h1, ..., out]`` the fused MLP kernels take (constants mirror csrc/mlp.hip)."""
    L = len(dims) - 1
    params = sum(dims[i + 1] * dims[i] + dims[i + 1] for i in range(L))
    return 1 <= L <= MLP_MAX_LAYERS and all(1 <= d <= MLP_MAX_WIDTH for d in dims) and params + 4 <= _MLP_LDS_FLOATS


def _ptr_array(tensors):
    import ctypes as C
    return (C.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def mlp_fwd(x, weights, biases, relu_last=False, save=True):
    """Whole Linear(+ReLU) stack in one launch.  ``x`` fp32 ``[B, in]``; ``weights[l]`` fp32 ``[out_l, in_l]``
    contiguous, ``biases[l]`` fp32 or ``None``.  Returns ``(out [B, out], acts [B, sum(in_l)] or None)``."""
    import ctypes as C
    lib = _lib.require()
    dims = [weights[0].shape[1]] + [w.shape[0] for w in weights]
    if x.dtype != torch.float32 or not x.is_contiguous() or x.shape[1] != dims[0]:
        raise ValueError("mlp_fwd: x must be a contiguous fp32 [B, in] tensor")
    if not mlp_supported(dims) or any(w.dtype != torch.float32 or not w.is_contiguous() for w in weights):
        raise ValueError("mlp_fwd: unsupported layer stack (see mlp_supported) or non-fp32 / non-contiguous weights")
    B = x.shape[0]
    out = torch.empty((B, dims[-1]), dtype=torch.float32, device=x.device)
    acts = torch.empty((B, sum(dims[:-1])), dtype=torch.float32, device=x.device) if save else None
    check(lib.ddl_mlp_fwd(ptr(x), _ptr_array(weights), _ptr_array(biases), (C.c_int * len(dims))(*dims), len(weights),
                          int(relu_last), ptr(acts), ptr(out), B, stream_ptr()), "mlp_fwd")
    return out, acts


def mlp_bwd(dout, out, acts, weights, dweights, dbiases, relu_last=False, want_dx=False):
    """Backward of :func:`mlp_fwd`: ``dweights[l]`` / ``dbiases[l]`` (fp32, parameter layouts, or ``None``) are
    ACCUMULATED into; returns ``dx`` (``[B, in]``) when asked for."""
    import ctypes as C
    lib = _lib.require()
    dims = [weights[0].shape[1]] + [w.shape[0] for w in weights]
    B = dout.shape[0]
    if dout.dtype != torch.float32 or not dout.is_contiguous() or tuple(dout.shape) != (B, dims[-1]):
        raise ValueError("mlp_bwd: dout must be a contiguous fp32 [B, out] tensor")
    for g, w in zip(dweights, weights):
        if g is not None and (g.dtype != torch.float32 or not g.is_contiguous() or g.shape != w.shape):
            raise ValueError("mlp_bwd: weight-gradient buffers must match the weights (fp32, contiguous)")
    dx = torch.empty((B, dims[0]), dtype=torch.float32, device=dout.device) if want_dx else None
    check(lib.ddl_mlp_bwd(ptr(dout), ptr(out), ptr(acts), _ptr_array(weights), _ptr_array(dweights), _ptr_array(dbiases),
                          (C.c_int * len(dims))(*dims), len(weights), int(relu_last), ptr(dx), B, stream_ptr()), "mlp_bwd")
    return dx


# ----------------------------------------------------------------------------------------------- recurrent
def lstm_supported(hidden: int, inputs: int) -> bool:
    """Shapes with a compiled instance of the fused LSTM layer (constants mirrored from csrc/lstm.hip)."""
    return hidden in (64, 128) and inputs in (32, 64, 128)


def rows_gemm_nt(a: torch.Tensor, bt: torch.Tensor) -> torch.Tensor:
    """bf16 ``[M, N] = a[M, K] @ bt[N, K]^T`` for a huge ``M``, ``K % 32 == 0`` and ``N`` in 32 / 64 / 128: the LSTM
    input gradient ``dG @ W_ih`` (``csrc/lstm.hip``), ``bt`` being the weight's ``[in, 4H]`` transpose."""
    lib = _lib.require()
    _chk_act(a, "rows_gemm_nt")
    _chk_act(bt, "rows_gemm_nt")
    M, Kr = a.shape
    N, K2 = bt.shape
    if K2 != Kr or Kr % 32 or N not in (32, 64, 128) or not a.is_contiguous() or not bt.is_contiguous():
        raise ValueError("rows_gemm_nt: need contiguous a [M, K], bt [N, K] with K % 32 == 0 and N in (32, 64, 128)")
    out = torch.empty((M, N), dtype=BF16, device=a.device)
    check(lib.ddl_rows_gemm_nt(ptr(a), ptr(bt), ptr(out), M, Kr, N, stream_ptr()), "rows_gemm_nt")
    return out


def lstm_fwd(x, w_ih, w_hh, bias, h0=None, c0=None, save=True):
    """One LSTM layer, all T steps in one launch.  ``x`` bf16 ``[B,T,I]`` (innermost contiguous), ``w_ih`` bf16
    ``[4H,I]``, ``w_hh`` bf16 ``[4H,H]``, ``bias`` fp32 ``[4H]`` (= b_ih + b_hh), optional fp32/bf16 ``h0`` / ``c0``
    ``[B,H]``.  Returns ``(hseq bf16 [B,T+1,H], cseq fp32 [B,T+1,H], gates bf16 [B,T,4,H] or None)``; row 0 of
    ``hseq`` / ``cseq`` is the initial state, ``hseq[:, 1:]`` the output sequence."""
    lib = _lib.require()
    B, T, I = x.shape
    H = w_hh.shape[1]
    if x.dtype != BF16 or w_ih.dtype != BF16 or w_hh.dtype != BF16 or bias.dtype != torch.float32:
        raise TypeError("lstm_fwd: x / w_ih / w_hh must be bf16 and bias fp32")
    if x.stride(2) != 1:
        x = x.contiguous()
    if tuple(w_ih.shape) != (4 * H, I) or tuple(w_hh.shape) != (4 * H, H) or bias.numel() != 4 * H:
        raise ValueError("lstm_fwd: weight shapes do not match")
    if not lstm_supported(H, I):
        raise ValueError(f"lstm_fwd: no compiled instance for hidden={H}, input={I}")
    w_ih, w_hh, bias = w_ih.contiguous(), w_hh.contiguous(), bias.contiguous()
    hseq = torch.empty((B, T + 1, H), dtype=BF16, device=x.device)
    cseq = torch.empty((B, T + 1, H), dtype=torch.float32, device=x.device)
    if h0 is None:
        hseq[:, 0].zero_()
    else:
        hseq[:, 0].copy_(h0)
    if c0 is None:
        cseq[:, 0].zero_()
    else:
        cseq[:, 0].copy_(c0)
    gates = torch.empty((B, T, 4, H), dtype=BF16, device=x.device) if save else None
    check(lib.ddl_lstm_fwd(ptr(x), x.stride(0), x.stride(1), ptr(w_ih), ptr(w_hh), ptr(bias), ptr(hseq), ptr(cseq),
                           ptr(gates), B, T, I, H, int(save), stream_ptr()), "lstm_fwd")
    return hseq, cseq, gates


def lstm_bwd(dout, gates, cseq, w_hh_t, want_state_grads=False):
    """Reverse recurrence.  ``dout`` bf16 ``[B,T,H]`` (gradient of ``hseq[:, 1:]``), ``w_hh_t`` bf16 ``[H,4H]``.
    Returns ``(dG bf16 [B,T,4H], dh0, dc0)`` — pre-activation gate gradients in PyTorch's i, f, g, o row order."""
    lib = _lib.require()
    B, T, four, H = gates.shape
    if dout.dtype != BF16 or tuple(dout.shape) != (B, T, H):
        raise ValueError("lstm_bwd: dout must be bf16 [B,T,H]")
    if dout.stride(2) != 1 or dout.stride(0) % 4 or dout.stride(1) % 4:
        dout = dout.contiguous()
    if tuple(w_hh_t.shape) != (H, 4 * H) or w_hh_t.dtype != BF16 or not w_hh_t.is_contiguous():
        raise ValueError("lstm_bwd: w_hh_t must be a contiguous bf16 [H,4H] tensor")
    dG = torch.empty((B, T, 4 * H), dtype=BF16, device=dout.device)
    dh0 = torch.empty((B, H), dtype=torch.float32, device=dout.device) if want_state_grads else None
    dc0 = torch.empty((B, H), dtype=torch.float32, device=dout.device) if want_state_grads else None
    check(lib.ddl_lstm_bwd(ptr(dout), dout.stride(0), dout.stride(1), ptr(gates), ptr(cseq), ptr(w_hh_t), ptr(dG),
                           ptr(dh0), ptr(dc0), B, T, H, stream_ptr()), "lstm_bwd")
    return dG, dh0, dc0


# ----------------------------------------------------------------------------------------------- loss
def softmax_xent(logits, target, want_grad=True):
    """Returns ``(out[2] = (mean loss, top-1 hits), dlogits or None)``; target is int64 ``[B]`` or fp32 ``[B,V]``."""
    lib = _lib.require()
    logits = logits.contiguous()
    B, V = logits.shape
    if logits.dtype not in (BF16, torch.float32):
        raise TypeError("softmax_xent: logits must be bf16 or fp32")
    labels = probs = None
    if target.dtype == torch.int64 and target.dim() == 1:
        labels = target.contiguous()
    else:
        probs = target.to(torch.float32).contiguous()
        if probs.shape != logits.shape:
            raise ValueError("softmax_xent: probability targets must match the logits shape")
    out = torch.empty((2,), dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    check(lib.ddl_softmax_xent(ptr(logits), int(logits.dtype == BF16), ptr(labels), ptr(probs), ptr(dl), ptr(out), B, V,
                               stream_ptr()), "softmax_xent")
    return out, dl
