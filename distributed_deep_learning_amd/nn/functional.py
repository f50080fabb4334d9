"""Autograd functions that put the hand-written gfx950 kernels on torch's tape.

Each ``torch.autograd.Function`` here is one fused unit of the hot path (conv+BN+ReLU(+residual), conv+bias+ReLU,
BN+ReLU, pooling, softmax-cross-entropy, linear).  They only accept bf16 channels-last GPU tensors; the CPU
(plumbing / gloo) path never reaches them — the modules in :mod:`.modules` dispatch on the tensor's device.

Reference parity: these cover the implicit device ops of the reference's training step
(``/root/reference/src/pytorch/CNN/main.py:82-92``: conv/BN/ReLU/pool forward, ``loss.backward()``).
"""
from __future__ import annotations

import torch

from ..ops import kernels as K

BF16 = torch.bfloat16


def _bf16(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == BF16 else t.to(BF16)


def _wgrad(weight, dy, x, R, S, stride, pad, cin_real):
    """Filter gradient.  When the parameter's ``.grad`` is a view of a flat arena, the split-K atomics of the wgrad
    kernel accumulate STRAIGHT into it (no temporary, no zero-fill, no autograd add) and ``None`` is returned to
    autograd; the arena then tells the data-parallel reducer that this tensor's gradient is ready."""
    arena = getattr(weight, "_ddl_arena", None)
    if arena is not None and cin_real == x.shape[1] and arena.grad_is_arena_view(weight):
        K.conv2d_wgrad(dy, x, R, S, stride, pad, out=weight.grad)
        arena.notify_grad_ready(weight)
        return None
    dw = K.conv2d_wgrad(dy, x, R, S, stride, pad)
    if cin_real != dw.shape[1]:
        dw = dw[:, :cin_real]
    return dw


class ConvBNAct(torch.autograd.Function):
    """y = act(BN_train(conv(x, w)) [+ residual]) with BN statistics fused into the conv epilogue."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, w_krsc, w_crsk, running_mean, running_var, momentum, eps,
                stride, pad, relu, cin_real, slabs):
        x = K.nhwc(_bf16(x))
        y_conv, stats = K.conv2d_fwd(x, w_krsc, stride, pad, want_stats=True, slab=slabs[0] if slabs else None)
        N, _, P, Q = y_conv.shape
        vec = K.bn_finalize(stats, N * P * Q, gamma, beta, running_mean, running_var, momentum, eps)
        res = K.nhwc(_bf16(residual)) if residual is not None else None
        out, mask = K.bn_apply(y_conv, vec[2], vec[3], residual=res, relu=relu, want_mask=True)
        ctx.save_for_backward(x, y_conv, mask, vec, gamma, w_crsk)
        ctx.geom = (stride, pad, relu, residual is not None, w_krsc.shape[1], w_krsc.shape[2], cin_real)
        ctx.weight = weight
        ctx.slab = slabs[1] if slabs else None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y_conv, mask, vec, gamma, w_crsk = ctx.saved_tensors
        stride, pad, relu, has_res, R, S, cin_real = ctx.geom
        dout = K.nhwc(_bf16(dout))
        dconv, dres, dgamma, dbeta = K.bn_backward(dout, mask, y_conv, gamma, vec[0], vec[1], relu=relu,
                                                   want_dres=has_res, slab=ctx.slab)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = K.conv2d_dgrad(dconv, w_crsk, tuple(x.shape[2:]), stride, pad)
        dw = _wgrad(ctx.weight, dconv, x, R, S, stride, pad, cin_real)
        return (dx, dw, dgamma, dbeta, dres) + (None,) * 11


class ConvBiasAct(torch.autograd.Function):
    """y = act(conv(x, w) + b) with bias and ReLU fused into the conv epilogue (VGG / LeNet style blocks)."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_krsc, w_crsk, stride, pad, relu, cin_real):
        x = K.nhwc(_bf16(x))
        y, _ = K.conv2d_fwd(x, w_krsc, stride, pad, bias=bias, relu=relu)
        ctx.save_for_backward(x, y if relu else None, w_crsk)
        ctx.geom = (stride, pad, relu, bias is not None, w_krsc.shape[1], w_krsc.shape[2], cin_real)
        ctx.weight = weight
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, w_crsk = ctx.saved_tensors
        stride, pad, relu, has_bias, R, S, cin_real = ctx.geom
        dy = K.nhwc(_bf16(dy))
        if relu:
            dy = K.relu_bwd(dy, y)
        db = K.bias_grad(dy) if has_bias else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = K.conv2d_dgrad(dy, w_crsk, tuple(x.shape[2:]), stride, pad)
        dw = _wgrad(ctx.weight, dy, x, R, S, stride, pad, cin_real)
        return (dx, dw, db) + (None,) * 6


class BNAct(torch.autograd.Function):
    """Stand-alone training BatchNorm (+ residual)(+ ReLU) for inputs not produced by our conv (e.g. concat)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, momentum, eps, relu, slabs):
        x = K.nhwc(_bf16(x))
        N, _, H, W = x.shape
        slab = K.bn_stats(x, slab=slabs[0] if slabs else None)
        ctx.slab = slabs[1] if slabs else None
        vec = K.bn_finalize(slab, N * H * W, gamma, beta, running_mean, running_var, momentum, eps)
        res = K.nhwc(_bf16(residual)) if residual is not None else None
        out, mask = K.bn_apply(x, vec[2], vec[3], residual=res, relu=relu, want_mask=True)
        ctx.save_for_backward(x, mask, vec, gamma)
        ctx.flags = (relu, residual is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, mask, vec, gamma = ctx.saved_tensors
        relu, has_res = ctx.flags
        dout = K.nhwc(_bf16(dout))
        dx, dres, dgamma, dbeta = K.bn_backward(dout, mask, x, gamma, vec[0], vec[1], relu=relu, want_dres=has_res,
                                                slab=ctx.slab)
        return (dx, dgamma, dbeta, dres) + (None,) * 6


class AffineAct(torch.autograd.Function):
    """Inference-mode BN: y = act(x*scale + shift [+ residual]); no parameter gradients (frozen statistics)."""

    @staticmethod
    def forward(ctx, x, scale, shift, residual, relu):
        x = K.nhwc(_bf16(x))
        res = K.nhwc(_bf16(residual)) if residual is not None else None
        out = K.bn_apply(x, scale, shift, residual=res, relu=relu)
        ctx.save_for_backward(out, scale)
        ctx.flags = (relu, residual is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, scale = ctx.saved_tensors
        relu, has_res = ctx.flags
        g = K.nhwc(_bf16(dout))
        if relu:
            g = K.relu_bwd(g, out)
        dx = (g.float() * scale.view(1, -1, 1, 1)).to(BF16).contiguous(memory_format=torch.channels_last)
        return dx, None, None, (g if has_res else None), None


class MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad):
        x = K.nhwc(_bf16(x))
        y, idx = K.maxpool_fwd(x, k, stride, pad)
        ctx.save_for_backward(idx)
        ctx.geom = (tuple(x.shape[2:]), k, stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        hw, k, stride, pad = ctx.geom
        return K.maxpool_bwd(K.nhwc(_bf16(dy)), idx, hw, k, stride, pad), None, None, None


class AvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride):
        x = K.nhwc(_bf16(x))
        ctx.geom = (tuple(x.shape[2:]), k, stride)
        return K.avgpool_fwd(x, k, stride)

    @staticmethod
    def backward(ctx, dy):
        hw, k, stride = ctx.geom
        return K.avgpool_bwd(K.nhwc(_bf16(dy)), hw, k, stride), None, None


class GlobalAvgPool(torch.autograd.Function):
    """[N,C,H,W] -> [N,C]"""

    @staticmethod
    def forward(ctx, x):
        x = K.nhwc(_bf16(x))
        ctx.hw = tuple(x.shape[2:])
        return K.gap_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return K.gap_bwd(_bf16(dy), ctx.hw)


class ReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = K.relu_fwd(_bf16(x))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return K.relu_bwd(_bf16(dy), y)


class Linear(torch.autograd.Function):
    """bf16 GEMM on hipBLASLt (a plain library GEMM) against the cached bf16 copy of the fp32 master weight."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_bf16):
        x = _bf16(x)
        ctx.save_for_backward(x, w_bf16)
        ctx.has_bias = bias is not None
        y = x @ w_bf16.t()
        if bias is not None:
            y = y + bias.to(BF16)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_bf16 = ctx.saved_tensors
        dy = _bf16(dy)
        dx = dy @ w_bf16 if ctx.needs_input_grad[0] else None
        dw = (dy.t() @ x).float()
        db = dy.float().sum(0) if ctx.has_bias else None
        return dx, dw, db, None


class SoftmaxCrossEntropy(torch.autograd.Function):
    """Mean softmax-CE on logits; the gradient is produced in the forward launch and scaled in backward."""

    @staticmethod
    def forward(ctx, logits, target):
        out, dl = K.softmax_xent(logits, target, want_grad=True)
        ctx.save_for_backward(dl)
        ctx.mark_non_differentiable(out)
        loss = out[0].clone()
        ctx.stats = out
        return loss, out

    @staticmethod
    def backward(ctx, gloss, _gout):
        (dl,) = ctx.saved_tensors
        return dl * gloss.to(dl.dtype), None
