"""Recurrent layers.  GPU path: one launch per LSTM layer and direction of time (``csrc/lstm.hip`` — weights in
registers, h in LDS, c in registers, the whole time loop inside the kernel) instead of one library call per gate
GEMM and time step.

Replaces ``torch.nn.LSTM`` as used by the reference's LSTM workload
(``/root/reference/src/pytorch/LSTM/model.py:70-94``); parameter names and shapes are those of ``torch.nn.LSTM`` so
checkpoints are interchangeable.
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from ..ops import kernels as K
from ..ops import state as _state

BF16 = torch.bfloat16


def _mm_f32(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """bf16 x bf16 -> fp32 GEMM (weight gradients over B*T rows keep their low bits)."""
    try:
        return torch.mm(a, b, out_dtype=torch.float32)
    except (TypeError, RuntimeError):
        return torch.mm(a, b).float()


def _wgrad_rows(g2: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """``g2^T @ x2`` for ``g2 [M, K]``, ``x2 [M, C]`` (bf16, M = B*T huge, K and C small) -> fp32 ``[K, C]``.
    This tall-skinny reduction is exactly a 1x1 convolution's filter gradient over M "pixels": the split-M wgrad
    kernel (``csrc/conv_wgrad.hip``) runs it several times faster than the BLAS GEMM without split-K."""
    M, Kc = g2.shape
    Cc = x2.shape[1]
    if M < 4096 or Kc % 8 or Cc % 8:
        return _mm_f32(g2.t(), x2)
    dw = K.conv2d_wgrad(g2.view(M, Kc, 1, 1), x2.view(M, Cc, 1, 1), 1, 1, 1, 0)
    return dw.reshape(Kc, Cc)


class LSTMLayerFn(torch.autograd.Function):
    """``out[B,T,H] (bf16) = LSTM(x[B,T,I])`` with zero initial state, PyTorch gate order i, f, g, o."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, w_ih16, w_hh16):
        xb = x if x.dtype == BF16 else x.to(BF16)
        bias = (b_ih.detach().float() + b_hh.detach().float()) if b_ih is not None else \
            torch.zeros(w_hh.shape[0], dtype=torch.float32, device=x.device)
        need = any(ctx.needs_input_grad[:5])
        hseq, cseq, gates = K.lstm_fwd(xb, w_ih16, w_hh16, bias, save=need)
        if need:
            ctx.save_for_backward(xb, hseq, cseq, gates, w_ih16, w_hh16)
            ctx.x_dtype = x.dtype
            ctx.has_bias = b_ih is not None
        return hseq[:, 1:]

    @staticmethod
    def backward(ctx, dout):
        xb, hseq, cseq, gates, w_ih16, w_hh16 = ctx.saved_tensors
        B, T, I = xb.shape
        H = w_hh16.shape[1]
        if dout.dtype != BF16:
            dout = dout.to(BF16)
        dG, _, _ = K.lstm_bwd(dout, gates, cseq, w_hh16.t().contiguous())
        g2 = dG.view(B * T, 4 * H)
        dw_ih = dw_hh = db = dx = None
        if ctx.needs_input_grad[1]:
            dw_ih = _wgrad_rows(g2, xb.reshape(B * T, I))
        if ctx.needs_input_grad[2]:
            dw_hh = _wgrad_rows(g2, hseq[:, :T].reshape(B * T, H))
        if ctx.has_bias and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]):
            db = K.bias_grad(g2) if (4 * H) % 8 == 0 else g2.sum(0, dtype=torch.float32)   # one pass, fp32 sums
        if ctx.needs_input_grad[0]:
            dx = torch.mm(g2, w_ih16).view(B, T, I).to(ctx.x_dtype)
        return dx, dw_ih, dw_hh, db, db, None, None


class LSTM(tnn.LSTM):
    """Single-layer, unidirectional, batch-first LSTM (what the reference stacks).  Returns ``(out, (h_n, c_n))`` like
    ``torch.nn.LSTM``; on the GPU ``out`` / ``h_n`` are bf16 and ``c_n`` is ``None`` (the reference never reads it).
    Other configurations, an explicit initial state, CPU tensors and shapes without a compiled kernel instance run the
    stock ``torch.nn.LSTM``."""

    def __init__(self, input_size, hidden_size, **kwargs):
        kwargs.setdefault("batch_first", True)
        super().__init__(input_size, hidden_size, **kwargs)
        self._pack = None

    def native_ok(self, x: torch.Tensor) -> bool:
        return (x.is_cuda and x.dim() == 3 and self.num_layers == 1 and not self.bidirectional and self.batch_first
                and self.proj_size == 0 and K.lstm_supported(self.hidden_size, self.input_size))

    def _w16(self):
        wi, wh = self.weight_ih_l0, self.weight_hh_l0
        key = (wi._version, wh._version, _state.weights_epoch(), wi.data_ptr(), wh.data_ptr())
        if self._pack is None or self._pack[0] != key:
            self._pack = (key, wi.detach().to(BF16).contiguous(), wh.detach().to(BF16).contiguous())
        return self._pack[1], self._pack[2]

    def forward(self, x, hx=None):
        if hx is not None or not self.native_ok(x):
            return super().forward(x.to(self.weight_ih_l0.dtype) if x.is_floating_point() else x, hx)
        wi16, wh16 = self._w16()
        b_ih = self.bias_ih_l0 if self.bias else None
        b_hh = self.bias_hh_l0 if self.bias else None
        out = LSTMLayerFn.apply(x, self.weight_ih_l0, self.weight_hh_l0, b_ih, b_hh, wi16, wh16)
        return out, (out[:, -1].unsqueeze(0), None)
