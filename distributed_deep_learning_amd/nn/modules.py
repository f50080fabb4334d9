"""Layers: thin subclasses of the torch modules (identical parameter names / shapes, so ``state_dict`` is
interchangeable with stock PyTorch / torchvision) whose GPU forward runs the hand-written gfx950 kernels.

Device dispatch, not backend dispatch: a CPU tensor (the gloo plumbing path, unit tests) takes the plain
torch fp32 op; a GPU tensor takes the HIP kernels on bf16 channels-last data and *raises* if the kernel library
is missing.  Filters keep an fp32 master in channels-last (KRSC) memory and a cached bf16 KRSC / CRSK pair
that is refreshed whenever the master changes (``Parameter._version`` or :func:`ops.state.bump`).
"""
from __future__ import annotations

import os
import warnings

import torch
import torch.nn as tnn
import torch.nn.functional as TF

from ..ops import kernels as K
from ..ops import state as _state
from . import functional as F

CL = torch.channels_last
BF16 = torch.bfloat16


def _pad8(c: int) -> int:
    return (c + 7) // 8 * 8


def to_device_input(x: torch.Tensor) -> torch.Tensor:
    """Images -> what the GPU path wants: bf16, channels-last, channel count padded to a multiple of 8."""
    if not x.is_cuda:
        return x
    c = x.shape[1]
    cp = _pad8(c)
    if x.dtype == torch.float32 and x.is_contiguous():
        return K.nchw_to_nhwc_pad(x, cp)
    x = x.to(BF16)
    if cp != c:
        x = TF.pad(x, (0, 0, 0, 0, 0, cp - c))
    return x.contiguous(memory_format=CL)


class Conv2d(tnn.Conv2d):
    """2-D convolution; GPU path = implicit-GEMM MFMA kernels (fwd / dgrad / wgrad)."""

    def __init__(self, *args, pad_out: bool = False, **kwargs):
        """``pad_out=True``: on the GPU an output-channel count that is not a multiple of 8 (LeNet's 6) is padded
        PHYSICALLY — the result has ``pad8(out_channels)`` channels, the extra ones exactly zero — instead of taking
        the ATen fallback.  Meant for conv -> (ReLU / pool) -> conv chains: the next ``Conv2d`` accepts the padded
        tensor as is.  Parameters keep their logical shapes; CPU tensors are unaffected."""
        super().__init__(*args, **kwargs)
        self.weight.data = self.weight.data.contiguous(memory_format=CL)
        self.pad_out = bool(pad_out)
        self._pack = None
        self._bias_pack = None
        self._warned = False

    # ---- geometry the kernels support
    def _native_ok(self) -> bool:
        return (
            self.groups == 1
            and self.dilation == (1, 1)
            and self.stride[0] == self.stride[1]
            and self.padding[0] == self.padding[1]
            and not isinstance(self.padding, str)
            and self.padding_mode == "zeros"
            and (self.out_channels % 8 == 0 or self.pad_out)
        )

    def packed(self):
        """(bf16 KRSC, bf16 CRSK) with input channels zero-padded to a multiple of 8."""
        w = self.weight
        sh_t = getattr(w, "_ddl_shadow_t", None)
        if sh_t is not None and getattr(w, "_ddl_shadow_version", -1) == w._version:
            # arena-resident bf16 shadow + CRSK twin, kept current by the fused optimizer: nothing to launch
            return w._ddl_shadow.permute(0, 2, 3, 1), sh_t
        key = (w._version, _state.weights_epoch(), w.data_ptr(), w.device)
        if self._pack is not None and self._pack[0] == key:
            return self._pack[1], self._pack[2]
        with torch.no_grad():
            src = w.detach()
            k, cin = src.shape[0], src.shape[1]
            if cin % 8 != 0 or k % 8 != 0:
                src = TF.pad(src, (0, 0, 0, 0, 0, _pad8(cin) - cin, 0, _pad8(k) - k)).contiguous(memory_format=CL)
            elif not src.is_contiguous(memory_format=CL):
                src = src.contiguous(memory_format=CL)
            krsc, crsk = K.filter_repack(src)
        self._pack = (key, krsc, crsk)
        return krsc, crsk

    def _fallback(self, x):
        if not self._warned:
            warnings.warn(f"Conv2d{tuple(self.weight.shape)}: geometry not covered by the HIP kernels, using ATen",
                          stacklevel=3)
            self._warned = True
        y = TF.conv2d(x.to(self.weight.dtype), self.weight, self.bias, self.stride, self.padding, self.dilation,
                      self.groups)
        return y.to(BF16).contiguous(memory_format=CL) if x.is_cuda else y

    def forward(self, x, relu: bool = False):
        if not x.is_cuda:
            y = super().forward(x)
            return TF.relu(y) if relu else y
        if not self._native_ok():
            y = self._fallback(x)
            return TF.relu(y) if relu else y
        if x.shape[1] != _pad8(self.in_channels):
            x = to_device_input(x)
        krsc, crsk = self.packed()
        return F.ConvBiasAct.apply(x, self.weight, self.bias, krsc, crsk, self.stride[0], self.padding[0], relu,
                                   self.in_channels, self._kernel_bias())

    def _kernel_bias(self):
        """fp32 bias as the kernel wants it: zero-padded to the physical channel count when ``pad_out`` is active."""
        b = self.bias
        if b is None or self.out_channels % 8 == 0:
            return b
        key = (b._version, _state.weights_epoch(), b.data_ptr())
        if self._bias_pack is None or self._bias_pack[0] != key:
            self._bias_pack = (key, TF.pad(b.detach().float(), (0, _pad8(self.out_channels) - self.out_channels)))
        return self._bias_pack[1]


class BatchNorm2d(tnn.BatchNorm2d):
    """BatchNorm with optional fused residual add and ReLU: ``bn(x, residual=None, relu=False)``.

    GPU training keeps two persistent self-cleaning reduction slabs per layer (forward statistics, backward
    sums) and counts ``num_batches_tracked`` on the host, folding it into the buffer when the state is saved —
    no per-step memsets or counter kernels."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._slabs = None
        self._nbt_pending = 0

    def slabs(self, device):
        if self._slabs is None or self._slabs[0].device != device:
            self._slabs = (K.new_slab(self.num_features, device), K.new_slab(self.num_features, device))
        return self._slabs

    def _count_batch(self) -> None:
        self._nbt_pending += 1

    def flush_counters(self) -> None:
        if self._nbt_pending and self.num_batches_tracked is not None:
            self.num_batches_tracked += self._nbt_pending
        self._nbt_pending = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush_counters()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._nbt_pending = 0
        super()._load_from_state_dict(*args, **kwargs)

    def _eval_affine(self):
        with torch.no_grad():
            inv = torch.rsqrt(self.running_var.float() + self.eps)
            g = self.weight.float() if self.affine else torch.ones_like(inv)
            b = self.bias.float() if self.affine else torch.zeros_like(inv)
            scale = g * inv
            shift = b - self.running_mean.float() * scale
        return scale, shift

    def _momentum(self) -> float:
        if self.momentum is None:      # cumulative moving average (called after _count_batch)
            return 1.0 / float(self.num_batches_tracked.item() + self._nbt_pending)
        return self.momentum

    def forward(self, x, residual=None, relu: bool = False):
        use_batch_stats = self.training or not self.track_running_stats
        if (not x.is_cuda) or x.shape[1] % 8 != 0 or not self.affine:
            y = super().forward(x.float() if x.is_cuda else x)
            if residual is not None:
                y = y + residual
            if relu:
                y = TF.relu(y)
            return y.to(BF16).contiguous(memory_format=CL) if x.is_cuda else y
        if use_batch_stats:
            if self.training and self.track_running_stats:
                self._count_batch()
            return F.BNAct.apply(x, self.weight, self.bias, residual,
                                 self.running_mean if self.track_running_stats else None,
                                 self.running_var if self.track_running_stats else None,
                                 self._momentum(), self.eps, relu, self.slabs(x.device))
        scale, shift = self._eval_affine()
        return F.AffineAct.apply(x, scale, shift, residual, relu)


def conv_bn_act(conv: Conv2d, bn: BatchNorm2d, x, relu: bool = True, residual=None):
    """conv -> BN -> (+residual) -> ReLU as ONE fused unit on the GPU (BN statistics come out of the conv
    epilogue, no separate statistics pass); plain composition on the CPU."""
    if (x.is_cuda and bn.training and bn.track_running_stats and bn.affine and conv.bias is None
            and conv._native_ok()):
        if x.shape[1] != _pad8(conv.in_channels):
            x = to_device_input(x)
        krsc, crsk = conv.packed()
        bn._count_batch()
        return F.ConvBNAct.apply(x, conv.weight, bn.weight, bn.bias, residual, krsc, crsk, bn.running_mean,
                                 bn.running_var, bn._momentum(), bn.eps, conv.stride[0], conv.padding[0], relu,
                                 conv.in_channels, bn.slabs(x.device))
    return bn(conv(x), residual=residual, relu=relu)


_S2D_STEM = os.environ.get("DDL_S2D_STEM", "1") != "0"


def _s2d_stem_ok(conv: Conv2d, x) -> bool:
    return (_S2D_STEM and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 and conv.in_channels == 3
            and conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3)
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and not x.requires_grad and x.is_contiguous())


def conv_bn_relu_pool(conv: Conv2d, bn: BatchNorm2d, pool, x):
    """conv -> BN -> ReLU -> max-pool (the ResNet / DenseNet stem).  In training on the GPU this is ONE fused unit
    whose full-resolution activation and gradient are never written (:class:`functional.ConvBNReluPool`)."""
    k, s, p = getattr(pool, "kernel_size", None), getattr(pool, "stride", None), getattr(pool, "padding", None)
    simple = (isinstance(pool, tnn.MaxPool2d) and all(isinstance(v, int) for v in (k, s, p)) and pool.dilation == 1
              and not pool.ceil_mode and not pool.return_indices and 2 * p <= k and k * k < 255)
    if (simple and x.is_cuda and torch.is_grad_enabled() and bn.training and bn.track_running_stats and bn.affine
            and conv.bias is None and conv._native_ok() and conv.out_channels % 8 == 0):
        if _s2d_stem_ok(conv, x):
            # 7x7/2 stem on raw fp32 images: space-to-depth input + regrouped 4x4/1 filter (35 % shorter reduction
            # forward, 4 filter rows of 64 columns instead of 7 x 56 in the filter gradient)
            w2 = K.s2d_filter(conv.weight)
            krsc = w2.detach().permute(0, 2, 3, 1).to(BF16).contiguous()
            bn._count_batch()
            return F.ConvBNReluPool.apply(K.nchw_to_s2d(x), w2, bn.weight, bn.bias, krsc, None, bn.running_mean,
                                          bn.running_var, bn._momentum(), bn.eps, 1, 0, 16, bn.slabs(x.device),
                                          (k, s, p))
        if x.shape[1] != _pad8(conv.in_channels):
            x = to_device_input(x)
        krsc, crsk = conv.packed()
        bn._count_batch()
        return F.ConvBNReluPool.apply(x, conv.weight, bn.weight, bn.bias, krsc, crsk, bn.running_mean, bn.running_var,
                                      bn._momentum(), bn.eps, conv.stride[0], conv.padding[0], conv.in_channels,
                                      bn.slabs(x.device), (k, s, p))
    return pool(conv_bn_act(conv, bn, x, relu=True))


def conv_relu_chain(units, x):
    """Run ``units`` = ``[(Conv2d, MaxPool2d or None), ...]`` (conv + bias + ReLU, optionally followed by a max-pool)
    as one fused autograd node on the GPU (:class:`functional.ConvReluChain`); layer by layer elsewhere."""
    def pool_geom(p):
        k, s, pd = p.kernel_size, p.stride, p.padding
        ok = (isinstance(p, tnn.MaxPool2d) and all(isinstance(v, int) for v in (k, s, pd)) and p.dilation == 1
              and not p.ceil_mode and not p.return_indices)
        return (k, s, pd) if ok else None

    geoms = [None if p is None else pool_geom(p) for _, p in units]
    fused = (x.is_cuda and torch.is_grad_enabled() and len(units) >= 2
             and all(c.bias is not None and c._native_ok() and c.out_channels % 8 == 0 for c, _ in units)
             and all(c.in_channels % 8 == 0 for c, _ in units[1:])
             and all(g is not None for (_, p), g in zip(units, geoms) if p is not None))
    if fused:
        if x.shape[1] != _pad8(units[0][0].in_channels):
            x = to_device_input(x)
        params = []
        for c, _ in units:
            params += [c.weight, c.bias]
        return F.ConvReluChain.apply(x, tuple((c, g) for (c, _), g in zip(units, geoms)), *params)
    for c, p in units:
        x = c(x, relu=True)
        if p is not None:
            x = p(x)
    return x


def residual_block(units, ds, x):
    """Run a residual block — ``units``: main-path (conv, bn) pairs, ``ds``: optional projection (conv, bn) — as one
    fused autograd node on the GPU (see :class:`functional.ResidualBlock`); layer-by-layer elsewhere."""
    mods = ([ds] if ds is not None else []) + list(units)
    fused = (x.is_cuda and torch.is_grad_enabled() and len(units) >= 2 and x.shape[1] % 8 == 0
             and all(c.bias is None and c._native_ok() and c.in_channels % 8 == 0 and b.training
                     and b.track_running_stats and b.affine for c, b in mods))
    if fused:
        params = []
        for c, b in mods:
            params += [c.weight, b.weight, b.bias]
        return F.ResidualBlock.apply(x, tuple(units), ds, *params)
    identity = x if ds is None else conv_bn_act(ds[0], ds[1], x, relu=False)
    h = x
    for i, (c, b) in enumerate(units):
        h = conv_bn_act(c, b, h, relu=True, residual=identity if i == len(units) - 1 else None)
    return h


_DENSE_BUFFER = os.environ.get("DDL_DENSE_BUFFER", "1") != "0"      # 0: concatenate layer by layer (A/B, debugging)


def dense_block(layers, x):
    """Run a DenseNet block — ``layers``: ``(norm1, conv1, norm2, conv2)`` per bottleneck layer, each computing
    ``conv2(relu(norm2(conv1(relu(norm1(concat of everything before))))))`` — and return the concatenation of the
    input and every layer's output.  On the GPU the block lives in ONE feature buffer that the layers read and extend
    in place (:class:`functional.DenseBlock`); elsewhere the features are concatenated layer by layer."""
    mods_ok = all(c.bias is None and c._native_ok() and c.in_channels % 8 == 0 and c.out_channels % 8 == 0
                  and b.affine and b.track_running_stats
                  for n1, c1, n2, c2 in layers for c, b in ((c1, n1), (c2, n2)))
    if x.is_cuda and mods_ok and x.shape[1] % 8 == 0 and layers and _DENSE_BUFFER:
        training = all(b.training for n1, _, n2, _ in layers for b in (n1, n2))
        if training and torch.is_grad_enabled():
            params = []
            for n1, c1, n2, c2 in layers:
                params += [n1.weight, n1.bias, c1.weight, n2.weight, n2.bias, c2.weight]
            return F.DenseBlock.apply(x, tuple(layers), *params)
        if not torch.is_grad_enabled() and not any(b.training for n1, _, n2, _ in layers for b in (n1, n2)):
            return F.dense_block_eval(x, layers)
    feats = [x]
    for n1, c1, n2, c2 in layers:
        h = torch.cat(feats, 1) if len(feats) > 1 else feats[0]
        h = n1(h, relu=True)
        h = conv_bn_act(c1, n2, h, relu=True)
        feats.append(c2(h))
    return torch.cat(feats, 1)


class ReLU(tnn.Module):
    def __init__(self, inplace: bool = False):
        super().__init__()

    def forward(self, x):
        if x.is_cuda and x.dtype == BF16 and x.numel() % 8 == 0:
            return F.ReLU.apply(x)
        return TF.relu(x)


class MaxPool2d(tnn.MaxPool2d):
    def forward(self, x):
        k, s, p = self.kernel_size, self.stride, self.padding
        simple = all(isinstance(v, int) for v in (k, s, p)) and self.dilation == 1 and not self.ceil_mode
        if x.is_cuda and simple and x.shape[1] % 8 == 0 and not self.return_indices:
            return F.MaxPool.apply(x, k, s, p)
        return super().forward(x)


class AvgPool2d(tnn.AvgPool2d):
    def forward(self, x):
        k, s = self.kernel_size, self.stride
        simple = isinstance(k, int) and isinstance(s, int) and self.padding == 0 and s >= k and not self.ceil_mode
        if x.is_cuda and simple and x.shape[1] % 8 == 0:
            return F.AvgPool.apply(x, k, s)
        return super().forward(x)


class GlobalAvgPool2d(tnn.Module):
    """``AdaptiveAvgPool2d(1)`` + ``flatten``: [N,C,H,W] -> [N,C]."""

    def forward(self, x):
        if x.is_cuda and x.shape[1] % 8 == 0:
            return F.GlobalAvgPool.apply(x)
        return x.mean((2, 3))


class Linear(tnn.Linear):
    """Fully connected layer.  GPU path: :class:`functional.LinearAct` — forward, input-gradient and weight-gradient
    GEMMs on the hand-written split-M MFMA kernel, bias / ReLU / dropout fused around them:
    ``lin(x, relu=True, dropout=0.5)``.  Feature counts that are not multiples of 8 are zero-padded internally."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._pack = None
        self._bias_pack = None

    def packed(self):
        """(bf16 ``[out_p, in_p]``, bf16 ``[in_p, out_p]``): the arena's shadow pair, or a cached private copy."""
        w = self.weight
        sh_t = getattr(w, "_ddl_shadow_t", None)
        if sh_t is not None and getattr(w, "_ddl_shadow_version", -1) == w._version:
            return w._ddl_shadow, sh_t
        key = (w._version, _state.weights_epoch(), w.data_ptr(), w.device)
        if self._pack is None or self._pack[0] != key:
            with torch.no_grad():
                src = w.detach()
                o, i = src.shape
                if o % 8 or i % 8:
                    src = TF.pad(src, (0, _pad8(i) - i, 0, _pad8(o) - o))
                w16 = src.to(BF16).contiguous()
                self._pack = (key, w16, K.transpose_pad(w16, w16.shape[0]))
        return self._pack[1], self._pack[2]

    def _kernel_bias(self):
        b = self.bias
        if b is None or self.out_features % 8 == 0:
            return b
        key = (b._version, _state.weights_epoch(), b.data_ptr())
        if self._bias_pack is None or self._bias_pack[0] != key:
            self._bias_pack = (key, TF.pad(b.detach().float(), (0, _pad8(self.out_features) - self.out_features)))
        return self._bias_pack[1]

    def forward(self, x, relu: bool = False, dropout: float = 0.0):
        if not x.is_cuda:
            y = super().forward(x)
            if relu:
                y = TF.relu(y)
            return TF.dropout(y, dropout, self.training) if dropout > 0.0 else y
        if x.dim() != 2:
            x = x.reshape(-1, x.shape[-1]) if x.dim() == 3 else x.reshape(x.shape[0], -1)
        w16, w16_t = self.packed()
        return F.LinearAct.apply(x, self.weight, self.bias, w16, w16_t, self._kernel_bias(), relu,
                                 float(dropout) if self.training else 0.0)


class L1Loss(tnn.Module):
    """Mean absolute error (the reference's LSTM criterion); GPU path = one fused forward+gradient launch."""

    last_correct = None

    def forward(self, out, target):
        if out.is_cuda:
            return F.L1Loss.apply(out, target)
        return TF.l1_loss(out.float(), target.float())


class Flatten(tnn.Module):
    """Flatten in logical NCHW order (so classifier weights stay state_dict-compatible with NCHW models)."""

    def forward(self, x):
        return x.reshape(x.shape[0], -1)


class SoftmaxCrossEntropyLoss(tnn.Module):
    """Mean softmax cross-entropy on **logits**.  Targets: int64 class ids or fp32 probability rows (the
    reference feeds one-hot floats, ``/root/reference/src/pytorch/CNN/dataset.py:108``).  ``forward`` returns
    the loss; ``last_correct`` holds the top-1 hit count of the same launch (a device tensor, no host sync)."""

    def __init__(self):
        super().__init__()
        self.last_correct = None

    def forward(self, logits, target):
        if logits.is_cuda:
            loss, out = F.SoftmaxCrossEntropy.apply(logits, target)
            self.last_correct = out[1]
            return loss
        lg = logits.float()
        loss = TF.cross_entropy(lg, target)
        with torch.no_grad():
            t = target if target.dim() == 1 else target.argmax(1)
            self.last_correct = (lg.argmax(1) == t).sum().float()
        return loss
