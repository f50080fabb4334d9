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


def _wgrad_rows(g2: torch.Tensor, x2: torch.Tensor, weight: torch.Tensor):
    """``g2^T @ x2`` for ``g2 [M, K]``, ``x2 [M, C]`` (bf16, M = B*T huge, K and C small) in fp32 ``[K, C]``.  This
    tall-skinny reduction is exactly a 1x1 convolution's filter gradient over M "pixels": the split-M MFMA kernel
    (``ops.kernels.gemm_tn``) spreads it over the whole chip.  A parameter that lives in an arena gets the product
    accumulated straight into its gradient there (returns ``None`` for autograd) - on the CURRENT stream: forking these
    15-80 us GEMMs to the wgrad side stream was measured on the reference's LSTM workload and costs more than it hides
    (hipGraph step 0.480 -> 0.543 ms at batch 32, 0.997 -> 1.046 ms at batch 4096: ``bench/ref_workloads.py``)."""
    arena = getattr(weight, "_ddl_arena", None)
    if arena is not None and arena.grad_is_arena_view(weight) and weight.grad.is_contiguous():
        K.gemm_tn(g2, x2, weight.grad)
        arena.notify_grad_ready(weight)
        return None
    dw = torch.zeros((g2.shape[1], x2.shape[1]), dtype=torch.float32, device=g2.device)
    return K.gemm_tn(g2, x2, dw)


class LSTMLayerFn(torch.autograd.Function):
    """``out[B,T,H] (bf16) = LSTM(x[B,T,I])`` with zero initial state, PyTorch gate order i, f, g, o.  ``w_*16`` are
    the bf16 weights, ``w_*16_t`` their ``[in, 4H]`` transposes (the arena's shadow pairs when the parameters live
    there).  Backward: the fused recurrent kernel for the gate gradients, then the two weight gradients on the
    split-M MFMA kernel (straight into the arena) and the input gradient on the row GEMM of
    ``csrc/lstm.hip`` - no library GEMM."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, w_ih16, w_hh16, w_ih16_t, w_hh16_t):
        xb = x if x.dtype == BF16 else x.to(BF16)
        bias = (b_ih.detach().float() + b_hh.detach().float()) if b_ih is not None else \
            torch.zeros(w_hh.shape[0], dtype=torch.float32, device=x.device)
        need = any(ctx.needs_input_grad[:5])
        hseq, cseq, gates = K.lstm_fwd(xb, w_ih16, w_hh16, bias, save=need)
        if need:
            ctx.save_for_backward(xb, hseq, cseq, gates, w_ih16_t, w_hh16_t)
            ctx.x_dtype = x.dtype
            ctx.has_bias = b_ih is not None
            ctx.weights = (w_ih, w_hh)
        return hseq[:, 1:]

    @staticmethod
    def backward(ctx, dout):
        xb, hseq, cseq, gates, w_ih16_t, w_hh16_t = ctx.saved_tensors
        w_ih, w_hh = ctx.weights
        B, T, I = xb.shape
        H = w_hh16_t.shape[0]
        if dout.dtype != BF16:
            dout = dout.to(BF16)
        dG, _, _ = K.lstm_bwd(dout, gates, cseq, w_hh16_t)
        g2 = dG.view(B * T, 4 * H)
        dw_ih = dw_hh = db = dx = None
        if ctx.needs_input_grad[1]:
            dw_ih = _wgrad_rows(g2, xb.reshape(B * T, I), w_ih)
        if ctx.needs_input_grad[2]:
            dw_hh = _wgrad_rows(g2, hseq[:, :T].reshape(B * T, H), w_hh)
        if ctx.has_bias and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]):
            db = K.bias_grad(g2)                                                  # one pass, fp32 sums
        if ctx.needs_input_grad[0]:
            dx = K.rows_gemm_nt(g2, w_ih16_t).view(B, T, I)
            if ctx.x_dtype != BF16:
                dx = dx.to(ctx.x_dtype)
        return dx, dw_ih, dw_hh, db, db, None, None, None, None


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
        """bf16 ``(w_ih [4H,I], w_hh [4H,H], w_ih^T [I,4H], w_hh^T [H,4H])``: the arena's shadow pairs (refreshed by
        the optimizer's own launch) or a private copy cached until the weights change."""
        wi, wh = self.weight_ih_l0, self.weight_hh_l0
        if all(getattr(w, "_ddl_shadow_t", None) is not None and getattr(w, "_ddl_shadow_version", -1) == w._version
               for w in (wi, wh)):
            return wi._ddl_shadow, wh._ddl_shadow, wi._ddl_shadow_t, wh._ddl_shadow_t      # noqa: SLF001
        key = (wi._version, wh._version, _state.weights_epoch(), wi.data_ptr(), wh.data_ptr())
        if self._pack is None or self._pack[0] != key:
            wi16, wh16 = K.cast_f32_bf16(wi.detach().float().contiguous()), K.cast_f32_bf16(wh.detach().float().contiguous())
            self._pack = (key, wi16, wh16, K.transpose_pad(wi16, wi16.shape[0]), K.transpose_pad(wh16, wh16.shape[0]))
        return self._pack[1:]

    def forward(self, x, hx=None):
        if hx is not None or not self.native_ok(x):
            return super().forward(x.to(self.weight_ih_l0.dtype) if x.is_floating_point() else x, hx)
        wi16, wh16, wi16_t, wh16_t = self._w16()
        b_ih = self.bias_ih_l0 if self.bias else None
        b_hh = self.bias_hh_l0 if self.bias else None
        out = LSTMLayerFn.apply(x, self.weight_ih_l0, self.weight_hh_l0, b_ih, b_hh, wi16, wh16, wi16_t, wh16_t)
        return out, (out[:, -1].unsqueeze(0), None)
