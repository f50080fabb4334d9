"""Autograd functions that put the hand-written gfx950 kernels on torch's tape.

Each ``torch.autograd.Function`` here is one fused unit of the hot path (conv+BN+ReLU(+residual), conv+bias+ReLU,
BN+ReLU, pooling, softmax-cross-entropy, linear).  They only accept bf16 channels-last GPU tensors; the CPU
(plumbing / gloo) path never reaches them — the modules in :mod:`.modules` dispatch on the tensor's device.

Reference parity: these cover the implicit device ops of the reference's training step
(``/root/reference/src/pytorch/CNN/main.py:82-92``: conv/BN/ReLU/pool forward, ``loss.backward()``).
"""
from __future__ import annotations

import os

import torch

from ..ops import kernels as K
from ..ops import streams

BF16 = torch.bfloat16


def _bf16(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == BF16 else t.to(BF16)


def _wgrad(weight, dy, x, R, S, stride, pad, cin_real, k_real=None):
    """Filter gradient.  When the parameter's ``.grad`` is a view of a flat arena, the split-K atomics of the wgrad
    kernel accumulate STRAIGHT into it (no temporary, no zero-fill, no autograd add) and ``None`` is returned to
    autograd; the arena then tells the data-parallel reducer that this tensor's gradient is ready."""
    arena = getattr(weight, "_ddl_arena", None)
    if k_real is not None and k_real != dy.shape[1]:          # physically padded output channels (Conv2d pad_out)
        return K.conv2d_wgrad(dy, x, R, S, stride, pad)[:k_real, :cin_real]
    if arena is not None and cin_real == x.shape[1] and arena.grad_is_arena_view(weight):
        with streams.fork(dy, x):          # side stream: overlaps the BN-backward / dgrad chain
            K.conv2d_wgrad(dy, x, R, S, stride, pad, out=weight.grad)
        arena.notify_grad_ready(weight)
        return None
    dw = K.conv2d_wgrad(dy, x, R, S, stride, pad)
    if cin_real != dw.shape[1]:
        dw = dw[:, :cin_real]
    return dw


def _bn_grad_out(gamma, beta):
    """(dgamma, dbeta) arena views to accumulate into, or None when the parameters are not arena-backed."""
    arena = getattr(gamma, "_ddl_arena", None)
    if (arena is not None and getattr(beta, "_ddl_arena", None) is arena and arena.grad_is_arena_view(gamma)
            and arena.grad_is_arena_view(beta)):
        return arena, (gamma.grad, beta.grad)
    return None, None


def _bn_bwd(dout, mask, x_bn, gamma_p, beta_p, gamma, mean, invstd, relu, want_dres, slab, presummed=False):
    """BN(+ReLU) backward; parameter gradients go straight into the arena when possible (then returned as None)."""
    arena, gout = _bn_grad_out(gamma_p, beta_p)
    dx, dres, dgamma, dbeta = K.bn_backward(dout, mask, x_bn, gamma, mean, invstd, relu=relu, want_dres=want_dres,
                                            slab=slab, grad_out=gout, presummed=presummed)
    if arena is not None:
        arena.notify_grad_ready(gamma_p)
        arena.notify_grad_ready(beta_p)
        dgamma = dbeta = None
    return dx, dres, dgamma, dbeta


_PAIR_BN = os.environ.get("DDL_BN_PAIR", "1") != "0"
# Backward launch order of a conv unit: filter gradient (side stream) BEFORE the data gradient (main stream).  Both read
# the same output gradient, which the BatchNorm-backward apply pass has just written: launched in this order they run
# side by side and the second reader finds the tensor in the caches (ResNet-50, batch 512: 38.07 -> 37.40 ms; deferring
# the filter gradient by one unit instead: 37.55-37.75 ms; by two: 37.9 ms; re-measured with round 3's halo filter-gradient
# kernels at batch 1000: 68.7 ms as is, 69.1 / 69.2 / 69.1 ms deferred by one / two / three units).  DDL_WGRAD_FIRST=0: data gradient first.
_WGRAD_FIRST = os.environ.get("DDL_WGRAD_FIRST", "1") != "0"
# conv + bias + ReLU units (VGG, LeNet): measured the other way round (VGG-16, batch 128: 17.8 ms data gradient first,
# 18.07 ms filter gradient first - the gated dgrad's epilogue also feeds the NEXT unit's bias gradient), so they keep it
_WGRAD_FIRST_PLAIN = os.environ.get("DDL_WGRAD_FIRST_PLAIN", "0") != "0"
# forward: projection-shortcut conv on the side stream, beside the main path's first units (A/B switch)
_FORK_SHORTCUT = os.environ.get("DDL_FORK_SHORTCUT", "1") != "0"
# (BN-backward sums taken in the producing dgrad's epilogue - skipping the reduce pass of the in-block BatchNorms - were
#  built in round 2 and measured slower on ResNet-50, 13.41k vs 13.49-13.54k img/s at batch 512: docs/STATUS.md.  The
#  option is gone; the dgrad epilogue's sums hook itself lives on in the conv + bias + ReLU chains of VGG below.)


def _bn_bwd_pair(dout, mask, saved, metas, ia, ib):
    """BN backward of units ``ia`` and ``ib`` of a residual block, both fed by ``dout`` behind ``mask``.
    Returns ``(dconv, dgamma, dbeta)`` per unit (gradients ``None`` when they went straight into the arena)."""
    _, ya, _, va, _ = saved[5 * ia:5 * ia + 5]
    _, yb, _, vb, _ = saved[5 * ib:5 * ib + 5]
    bna, bnb = metas[ia][1], metas[ib][1]
    ra, rb = K.vec_rows(va), K.vec_rows(vb)
    arena_a, gout_a = _bn_grad_out(bna.weight, bna.bias)
    arena_b, gout_b = _bn_grad_out(bnb.weight, bnb.bias)
    both = arena_a is not None and arena_b is not None
    dxa, dxb, ga, gb = K.bn_backward_pair(dout, mask, ya, yb, (bna.weight, ra[0], ra[1]), (bnb.weight, rb[0], rb[1]),
                                          relu=True, slabs=(metas[ia][4], metas[ib][4]),
                                          grad_outs=(gout_a, gout_b) if both else None)
    if both:
        for arena, bn in ((arena_a, bna), (arena_b, bnb)):
            arena.notify_grad_ready(bn.weight)
            arena.notify_grad_ready(bn.bias)
        return (dxa, None, None), (dxb, None, None)
    return (dxa, ga[0], ga[1]), (dxb, gb[0], gb[1])


class ConvBNAct(torch.autograd.Function):
    """y = act(BN_train(conv(x, w)) [+ residual]) with BN statistics fused into the conv epilogue."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, w_krsc, w_crsk, running_mean, running_var, momentum, eps,
                stride, pad, relu, cin_real, slabs):
        x = K.nhwc(_bf16(x))
        y_conv, vec = K.conv2d_fwd_bn(x, w_krsc, stride, pad, gamma, beta, running_mean, running_var, momentum, eps,
                                      slabs[0] if slabs else K.new_slab(w_krsc.shape[0], x.device))
        res = K.nhwc(_bf16(residual)) if residual is not None else None
        out, mask = K.bn_apply(y_conv, vec[2], vec[3], residual=res, relu=relu, want_mask=True)
        ctx.save_for_backward(x, y_conv, mask, vec, gamma, w_crsk)
        ctx.geom = (stride, pad, relu, residual is not None, w_krsc.shape[1], w_krsc.shape[2], cin_real)
        ctx.weight = weight
        ctx.bn_params = (gamma, beta)
        ctx.slab = slabs[1] if slabs else None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y_conv, mask, vec, gamma, w_crsk = ctx.saved_tensors
        stride, pad, relu, has_res, R, S, cin_real = ctx.geom
        dout = K.nhwc(_bf16(dout))
        dconv, dres, dgamma, dbeta = _bn_bwd(dout, mask, y_conv, ctx.bn_params[0], ctx.bn_params[1], gamma, vec[0],
                                             vec[1], relu, has_res, ctx.slab)
        dx = None
        dw = _wgrad(ctx.weight, dconv, x, R, S, stride, pad, cin_real) if _WGRAD_FIRST else None
        if ctx.needs_input_grad[0]:
            dx = K.conv2d_dgrad(dconv, w_crsk, tuple(x.shape[2:]), stride, pad)
        if not _WGRAD_FIRST:
            dw = _wgrad(ctx.weight, dconv, x, R, S, stride, pad, cin_real)
        return (dx, dw, dgamma, dbeta, dres) + (None,) * 11


class ConvBNReluPool(torch.autograd.Function):
    """Stem unit conv -> BN(train) -> ReLU -> max-pool with the un-pooled activation (and its gradient) never written:
    the pooling is folded into the BN apply pass forward and into both BN-backward passes (``csrc/bn.hip``)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, w_krsc, w_crsk, running_mean, running_var, momentum, eps, stride, pad,
                cin_real, slabs, pool):
        x = K.nhwc(_bf16(x))
        y_conv, vec = K.conv2d_fwd_bn(x, w_krsc, stride, pad, gamma, beta, running_mean, running_var, momentum, eps,
                                      slabs[0] if slabs else K.new_slab(w_krsc.shape[0], x.device))
        rows = K.vec_rows(vec)
        out, idx, aux = K.bn_relu_maxpool_fwd(y_conv, rows[2], rows[3], *pool, want_aux=True)
        ctx.save_for_backward(x, y_conv, idx, vec, gamma, w_crsk, *aux)
        ctx.geom = (stride, pad, w_krsc.shape[1], w_krsc.shape[2], cin_real, pool)
        ctx.weight = weight
        ctx.bn_params = (gamma, beta)
        ctx.slab = slabs[1] if slabs else None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y_conv, idx, vec, gamma, w_crsk, zmax, pmask = ctx.saved_tensors
        stride, pad, R, S, cin_real, pool = ctx.geom
        dout = K.nhwc(_bf16(dout))
        rows = K.vec_rows(vec)
        arena, gout = _bn_grad_out(*ctx.bn_params)
        dconv, dgamma, dbeta = K.bn_relu_maxpool_bwd(dout, idx, y_conv, gamma, rows[0], rows[1], *pool, slab=ctx.slab,
                                                     grad_out=gout, aux=(zmax, pmask))
        if arena is not None:
            arena.notify_grad_ready(ctx.bn_params[0])
            arena.notify_grad_ready(ctx.bn_params[1])
            dgamma = dbeta = None
        dx = None
        dw = _wgrad(ctx.weight, dconv, x, R, S, stride, pad, cin_real) if _WGRAD_FIRST else None
        if ctx.needs_input_grad[0]:
            dx = K.conv2d_dgrad(dconv, w_crsk, tuple(x.shape[2:]), stride, pad)
        if not _WGRAD_FIRST:
            dw = _wgrad(ctx.weight, dconv, x, R, S, stride, pad, cin_real)
        return (dx, dw, dgamma, dbeta) + (None,) * 11


class ResidualBlock(torch.autograd.Function):
    """A whole residual block (ResNet BasicBlock / Bottleneck) as ONE autograd node.

    ``units`` = main-path (conv, bn) pairs, every one followed by ReLU; the last one adds the shortcut before its
    ReLU.  ``ds`` = optional (conv, bn) projection shortcut.  Owning the block's backward lets the gradient of the
    block input be accumulated inside the first conv's dgrad epilogue (``addend``) instead of a separate add
    kernel, keeps every filter / BN gradient flowing straight into the arena, and replaces ~4 autograd nodes and
    their Python dispatch by one.  Parameters are passed as inputs (so non-arena setups still get gradients back).
    """

    @staticmethod
    def forward(ctx, x, units, ds, *params):
        x = K.nhwc(_bf16(x))
        saved, metas = [], []

        def run(conv, bn, inp, relu, residual, res_affine=None, apply=True):
            krsc, crsk = conv.packed()
            slabs = bn.slabs(inp.device)
            bn._count_batch()
            y_conv, vec = K.conv2d_fwd_bn(inp, krsc, conv.stride[0], conv.padding[0], bn.weight, bn.bias, bn.running_mean,
                                          bn.running_var, bn._momentum(), bn.eps, slabs[0])
            rows = K.vec_rows(vec)
            out = mask = None
            if apply:
                rs, rh = res_affine if res_affine is not None else (None, None)
                out, mask = K.bn_apply(y_conv, rows[2], rows[3], residual=residual, relu=relu, want_mask=True,
                                       res_scale=rs, res_shift=rh)
            saved.extend([inp, y_conv, mask, vec, crsk])
            metas.append((conv, bn, relu, residual is not None, slabs[1]))
            return out if apply else (y_conv, rows[2], rows[3])

        identity, ds_affine = x, None
        forked = False
        if ds is not None:
            # projection shortcut: only conv + statistics here; its BN affine is applied to the raw conv output inside
            # the last unit's apply pass, so the normalised shortcut is neither written nor read back.  Nothing but
            # that last apply pass needs it: it runs on the side stream, beside the main path's first units.
            if _FORK_SHORTCUT and x.is_cuda and len(units) > 1 and streams.enabled():
                with streams.fork(x):
                    identity, sc, sh = run(ds[0], ds[1], x, False, None, apply=False)
                forked = True
            else:
                identity, sc, sh = run(ds[0], ds[1], x, False, None, apply=False)
            ds_affine = (sc, sh)
        h = x
        for i, (conv, bn) in enumerate(units):
            last = i == len(units) - 1
            if last and forked:
                streams.join()
            h = run(conv, bn, h, True, identity if last else None, ds_affine if last else None)
        ctx.save_for_backward(*saved)
        ctx.metas = metas
        ctx.has_ds = ds is not None
        ctx.nparams = len(params)
        return h

    @staticmethod
    def backward(ctx, dout):
        saved = ctx.saved_tensors
        metas = ctx.metas
        grads = {}

        def back(idx, dy, need_dx, addend=None, addend_mask=None, relu_mask=None, want_dres=False, addend_step=1,
                 compact_dx=False, bn_done=None):
            """BN(+ReLU) backward -> dgrad (+ addend) -> wgrad of unit ``idx``.  ``relu_mask`` overrides the unit's
            own activation: the unit then sees ``dy * mask`` (used to push the block-output ReLU into the shortcut)."""
            inp, y_conv, mask, vec, crsk = saved[5 * idx:5 * idx + 5]
            conv, bn, relu, _has_res, slab = metas[idx]
            if relu_mask is not None:
                relu, mask = True, relu_mask
            rows = K.vec_rows(vec)
            if bn_done is not None:                  # BN backward already ran (paired with another unit's)
                dconv, dgamma, dbeta = bn_done
                dres = None
            else:
                dconv, dres, dgamma, dbeta = _bn_bwd(dy, mask, y_conv, bn.weight, bn.bias, bn.weight, rows[0], rows[1],
                                                     relu, want_dres, slab)
            dx = None
            R, S = conv.kernel_size
            dw = (_wgrad(conv.weight, dconv, inp, R, S, conv.stride[0], conv.padding[0], conv.in_channels)
                  if _WGRAD_FIRST else None)
            if need_dx and compact_dx:
                # strided 1x1 conv: its input gradient is non-zero only on the stride grid -> compute just that grid
                # (a plain stride-1 dgrad over the output pixels); the consumer adds it with ``addend_step``
                dx = K.conv2d_dgrad(dconv, crsk, tuple(dconv.shape[2:]), 1, 0)
            elif need_dx:
                dx = K.conv2d_dgrad(dconv, crsk, tuple(inp.shape[2:]), conv.stride[0], conv.padding[0], addend=addend,
                                    addend_mask=addend_mask, addend_step=addend_step)
            if not _WGRAD_FIRST:
                dw = _wgrad(conv.weight, dconv, inp, R, S, conv.stride[0], conv.padding[0], conv.in_channels)
            grads[idx] = (dw, dgamma, dbeta)
            return dx, dres

        dout = K.nhwc(_bf16(dout))
        first_main = 1 if ctx.has_ds else 0
        n_units = len(metas)
        need_x = ctx.needs_input_grad[0]
        last = n_units - 1
        out_mask = saved[5 * last + 2]                # ReLU mask of the block output
        if last > first_main:
            # The shortcut's gradient is dout * relu'(out).  It is never written: the projection's BN backward and the
            # first conv's dgrad epilogue apply the block-output mask to ``dout`` themselves.
            ds_done = None
            if ctx.has_ds and _PAIR_BN:
                # last main BN and shortcut BN see the same gradient behind the same mask: one pair of passes
                last_done, ds_done = _bn_bwd_pair(dout, out_mask, saved, metas, last, 0)
            if ds_done is not None:
                d, _ = back(last, dout, True, bn_done=last_done)
            else:
                d, _ = back(last, dout, True)
            for idx in range(last - 1, first_main, -1):
                d, _ = back(idx, d, True)
            if ctx.has_ds:
                dsc = metas[0][0]
                step = dsc.stride[0]
                fconv = metas[first_main][0]
                compact = (step > 1 and dsc.kernel_size == (1, 1) and dsc.padding[0] == 0
                           and fconv.stride[0] == 1)          # first main conv writes every input pixel in one launch
                # (issuing this dgrad early on the side stream, behind an event, was measured: no gain over the forward fork)
                dshort, _ = back(0, dout, need_x, relu_mask=out_mask, compact_dx=compact, bn_done=ds_done)
                dx, _ = back(first_main, d, need_x, addend=dshort if need_x else None,
                             addend_step=step if compact else 1)
            else:
                dx, _ = back(first_main, d, need_x, addend=dout if need_x else None,
                             addend_mask=out_mask if need_x else None)
        else:                                         # degenerate: the only main unit carries the residual
            dx, r = back(first_main, dout, need_x, want_dres=True)
            if ctx.has_ds:
                dshort, _ = back(0, r, need_x)
                r = dshort
            if need_x and r is not None:
                dx = K.add(dx, r)
        out = []
        for idx in range(n_units):
            out.extend(grads[idx])
        # parameter order given to apply(): ds first (if any), then the main units — same as metas
        return (dx, None, None, *out)


def _aux_slab(mod, channels, device):
    """A persistent self-cleaning statistics slab hung on a module that has no BatchNorm of its own."""
    slab = getattr(mod, "_ddl_aux_slab", None)
    if slab is None or slab.device != device or slab.shape[-1] != channels:
        slab = K.new_slab(channels, device)
        mod._ddl_aux_slab = slab
    return slab


class DenseBlock(torch.autograd.Function):
    """A whole DenseNet block as ONE autograd node working on ONE feature buffer ``[N, c0 + L*growth, H, W]``.

    The reference concatenates all earlier feature maps again for every layer
    (``/root/reference/src/pytorch/CNN/model.py:43-47,75-83``): O(L^2) copies forward, O(L^2) slice-adds backward.
    Here every layer is  BN1+ReLU (reads the buffer's channel prefix in place) -> conv1x1 -> BN2+ReLU -> conv3x3 whose
    epilogue writes the ``growth`` new channels straight into their window of the buffer.  Batch statistics of a
    channel are taken once, by the conv that produced it, and shared by every later BN1 through a per-block totals
    table (``csrc/bn.hip``).  Backward walks the layers in reverse over one gradient buffer: BN1's input gradient is
    accumulated into the prefix window, conv2's output gradient is the layer's own window.
    ``layers`` = ``[(norm1, conv1, norm2, conv2), ...]``; ``params`` = their parameters, six per layer."""

    @staticmethod
    def forward(ctx, x, layers, *params):
        x = K.nhwc(_bf16(x))
        N, c0, H, W = x.shape
        growth = [l[3].out_channels for l in layers]
        ctot = c0 + sum(growth)
        dev = x.device
        buf = torch.empty((N, ctot, H, W), dtype=BF16, device=dev, memory_format=torch.channels_last)
        tot = K.new_totals(ctot, dev)
        M = N * H * W
        K.copy_channels(x, K.ChannelWindow(buf, 0, c0))
        first = layers[0][0]
        K.slab_fold_totals(K.bn_stats(x, slab=first.slabs(dev)[0]), c0, tot, 0)
        saved, cin = [], c0
        for (n1, c1, n2, c2), g in zip(layers, growth):
            n1._count_batch()
            vec1 = K.bn_finalize_totals(tot, cin, M, n1.weight, n1.bias, n1.running_mean, n1.running_var, n1._momentum(),
                                        n1.eps)
            r1 = K.vec_rows(vec1)
            a1, m1 = K.bn_apply_window(K.ChannelWindow(buf, 0, cin), r1[2], r1[3], relu=True)
            k1, t1 = c1.packed()
            n2._count_batch()
            z1, vec2 = K.conv2d_fwd_bn(a1, k1, c1.stride[0], c1.padding[0], n2.weight, n2.bias, n2.running_mean,
                                       n2.running_var, n2._momentum(), n2.eps, n2.slabs(dev)[0])
            r2 = K.vec_rows(vec2)
            a2, m2 = K.bn_apply(z1, r2[2], r2[3], relu=True, want_mask=True)
            k2, t2 = c2.packed()
            st2 = K.conv2d_fwd_into(a2, k2, K.ChannelWindow(buf, cin, g), c2.stride[0], c2.padding[0],
                                    slab=_aux_slab(c2, g, dev))
            K.slab_fold_totals(st2, g, tot, cin)
            saved.extend([a1, m1, vec1, t1, z1, a2, m2, vec2, t2])
            cin += g
        ctx.save_for_backward(buf, *saved)
        ctx.layers, ctx.c0, ctx.growth = layers, c0, growth
        return buf

    @staticmethod
    def backward(ctx, dout):
        buf, *saved = ctx.saved_tensors
        layers, c0, growth = ctx.layers, ctx.c0, ctx.growth
        N, ctot, H, W = buf.shape
        # ONE gradient buffer for the block.  The upstream gradient is never modified: the last layer (first visited)
        # writes  dout[prefix] + its BN1 gradient  into the new buffer, every earlier layer accumulates in place.
        dout = K.nhwc(_bf16(dout))
        gbuf = torch.empty_like(dout, memory_format=torch.channels_last)
        src = dout
        grads = [None] * (6 * len(layers))
        cin = ctot
        for li in range(len(layers) - 1, -1, -1):
            n1, c1, n2, c2 = layers[li]
            g = growth[li]
            cin -= g
            a1, m1, vec1, t1, z1, a2, m2, vec2, t2 = saved[9 * li:9 * li + 9]
            r1, r2 = K.vec_rows(vec1), K.vec_rows(vec2)
            dy2 = torch.empty((N, g, H, W), dtype=BF16, device=buf.device, memory_format=torch.channels_last)
            K.copy_channels(K.ChannelWindow(src, cin, g), dy2)
            dw2 = (_wgrad(c2.weight, dy2, a2, *c2.kernel_size, c2.stride[0], c2.padding[0], c2.in_channels)
                   if _WGRAD_FIRST else None)
            da2 = K.conv2d_dgrad(dy2, t2, (H, W), c2.stride[0], c2.padding[0])
            if not _WGRAD_FIRST:
                dw2 = _wgrad(c2.weight, dy2, a2, *c2.kernel_size, c2.stride[0], c2.padding[0], c2.in_channels)
            dz1, _, dg2, db2 = _bn_bwd(da2, m2, z1, n2.weight, n2.bias, n2.weight, r2[0], r2[1], True, False,
                                       n2.slabs(buf.device)[1])
            dw1 = (_wgrad(c1.weight, dz1, a1, *c1.kernel_size, c1.stride[0], c1.padding[0], c1.in_channels)
                   if _WGRAD_FIRST else None)
            da1 = K.conv2d_dgrad(dz1, t1, (H, W), c1.stride[0], c1.padding[0])
            if not _WGRAD_FIRST:
                dw1 = _wgrad(c1.weight, dz1, a1, *c1.kernel_size, c1.stride[0], c1.padding[0], c1.in_channels)
            arena, gout = _bn_grad_out(n1.weight, n1.bias)
            dg1, db1 = K.bn_backward_window(da1, m1, K.ChannelWindow(buf, 0, cin), n1.weight, r1[0], r1[1],
                                            K.ChannelWindow(gbuf, 0, cin), add_to=K.ChannelWindow(src, 0, cin),
                                            relu=True, slab=n1.slabs(buf.device)[1], grad_out=gout)
            src = gbuf
            if arena is not None:
                arena.notify_grad_ready(n1.weight)
                arena.notify_grad_ready(n1.bias)
                dg1 = db1 = None
            grads[6 * li:6 * li + 6] = [dg1, db1, dw1, dg2, db2, dw2]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((N, c0, H, W), dtype=BF16, device=buf.device, memory_format=torch.channels_last)
            K.copy_channels(K.ChannelWindow(gbuf, 0, c0), dx)
        return (dx, None, *grads)


def dense_block_eval(x, layers):
    """Inference forward of a dense block on the same single buffer (running statistics, nothing saved)."""
    x = K.nhwc(_bf16(x))
    N, c0, H, W = x.shape
    growth = [l[3].out_channels for l in layers]
    buf = torch.empty((N, c0 + sum(growth), H, W), dtype=BF16, device=x.device,
                      memory_format=torch.channels_last)
    K.copy_channels(x, K.ChannelWindow(buf, 0, c0))
    cin = c0
    for (n1, c1, n2, c2), g in zip(layers, growth):
        s1, h1 = n1._eval_affine()
        a1, _ = K.bn_apply_window(K.ChannelWindow(buf, 0, cin), s1, h1, relu=True, want_mask=False)
        z1, _ = K.conv2d_fwd(a1, c1.packed()[0], c1.stride[0], c1.padding[0])
        s2, h2 = n2._eval_affine()
        a2 = K.bn_apply(z1, s2, h2, relu=True)
        K.conv2d_fwd_into(a2, c2.packed()[0], K.ChannelWindow(buf, cin, g), c2.stride[0], c2.padding[0],
                          want_stats=False)
        cin += g
    return buf


class ConvBiasAct(torch.autograd.Function):
    """y = act(conv(x, w) + b) with bias and ReLU fused into the conv epilogue (VGG / LeNet style blocks)."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_krsc, w_crsk, stride, pad, relu, cin_real, kernel_bias=None):
        x = K.nhwc(_bf16(x))
        y, _ = K.conv2d_fwd(x, w_krsc, stride, pad, bias=kernel_bias if kernel_bias is not None else bias, relu=relu)
        ctx.save_for_backward(x, y if relu else None, w_crsk)
        ctx.geom = (stride, pad, relu, bias is not None, w_krsc.shape[1], w_krsc.shape[2], cin_real)
        ctx.k_real = weight.shape[0]
        ctx.weight = weight
        ctx.bias = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, w_crsk = ctx.saved_tensors
        stride, pad, relu, has_bias, R, S, cin_real = ctx.geom
        dy = K.nhwc(_bf16(dy))
        db = None
        k_real = ctx.k_real
        padded = k_real != dy.shape[1]
        if relu and has_bias:                       # ReLU backward + bias gradient in one pass, straight into the arena
            bias = ctx.bias
            arena = getattr(bias, "_ddl_arena", None)
            if not padded and arena is not None and arena.grad_is_arena_view(bias):
                dy, _ = K.relu_bwd_bias_grad(dy, y, grad_out=bias.grad)
                arena.notify_grad_ready(bias)
            else:
                dy, db = K.relu_bwd_bias_grad(dy, y)
        else:
            if relu:
                dy = K.relu_bwd(dy, y)
            db = K.bias_grad(dy) if has_bias else None
        if padded and db is not None:
            db = db[:k_real]
        dx = None
        dw = _wgrad(ctx.weight, dy, x, R, S, stride, pad, cin_real, k_real) if _WGRAD_FIRST_PLAIN else None
        if ctx.needs_input_grad[0]:
            dx = K.conv2d_dgrad(dy, w_crsk, tuple(x.shape[2:]), stride, pad)
        if not _WGRAD_FIRST_PLAIN:
            dw = _wgrad(ctx.weight, dy, x, R, S, stride, pad, cin_real, k_real)
        return (dx, dw, db) + (None,) * 7


_RELU_GATE = os.environ.get("DDL_DGRAD_RELU_GATE", "1") != "0"


class ConvReluChain(torch.autograd.Function):
    """A run of ``conv + bias + ReLU (+ max-pool)`` units (VGG's feature extractor) as ONE autograd node.

    Owning the run's backward removes the ReLU-backward / bias-gradient pass (read dy, read y, write dx) of every unit
    but the last: unit ``i``'s dgrad differentiates ``a = relu(conv_{i-1} + b)`` (or its max-pool), which is its own
    input, so its epilogue stores ``dx * (a > 0)`` and sums it per channel — the gated gradient and the bias gradient
    of unit ``i-1`` — while the tile is on chip.  Gating the POOLED gradient is equivalent: a pooled value is > 0 iff
    its arg-max element is, and that is where max-pool backward routes the gradient.
    ``specs``: tuple of ``(conv_module, (k, stride, pad) or None)``; parameters follow as ``w0, b0, w1, b1, ...``.
    """

    @staticmethod
    def forward(ctx, x, specs, *params):
        h = K.nhwc(_bf16(x))
        saved = []
        for conv, pool in specs:
            krsc, crsk = conv.packed()
            y, _ = K.conv2d_fwd(h, krsc, conv.stride[0], conv.padding[0], bias=conv.bias, relu=True)
            idx = None
            nxt = y
            if pool is not None:
                nxt, idx = K.maxpool_fwd(y, *pool)
            saved.extend([h, y, crsk, idx])
            h = nxt
        ctx.save_for_backward(*saved)
        ctx.specs = specs
        return h

    @staticmethod
    def backward(ctx, dout):
        saved, specs = ctx.saved_tensors, ctx.specs
        n = len(specs)
        d = K.nhwc(_bf16(dout))
        grads = [None] * (2 * n)
        gated = False                       # d already carries this unit's ReLU gate, bias gradient already taken
        for i in range(n - 1, -1, -1):
            conv, pool = specs[i]
            x_in, y, crsk, idx = saved[4 * i:4 * i + 4]
            if pool is not None:
                d = K.maxpool_bwd(d, idx, tuple(y.shape[2:]), *pool)
            bias = conv.bias
            if not gated:
                arena = getattr(bias, "_ddl_arena", None)
                if arena is not None and arena.grad_is_arena_view(bias):
                    d, _ = K.relu_bwd_bias_grad(d, y, grad_out=bias.grad)
                    arena.notify_grad_ready(bias)
                else:
                    d, grads[2 * i + 1] = K.relu_bwd_bias_grad(d, y)
            R, S = conv.kernel_size
            stride, pad = conv.stride[0], conv.padding[0]
            if _WGRAD_FIRST_PLAIN:
                grads[2 * i] = _wgrad(conv.weight, d, x_in, R, S, stride, pad, conv.in_channels)
            dx = None
            gated = False
            if i > 0 or ctx.needs_input_grad[0]:
                if i > 0 and _RELU_GATE and K.dgrad_bn_sums_ok(d, crsk, stride, R, S):
                    pconv = specs[i - 1][0]
                    slab = getattr(conv, "_gate_slab", None)
                    if slab is None or slab.device != d.device:
                        slab = conv._gate_slab = K.new_slab(x_in.shape[1], d.device)
                    dx = K.conv2d_dgrad(d, crsk, tuple(x_in.shape[2:]), stride, pad, relu_gate=(x_in, slab))
                    pb = pconv.bias
                    arena = getattr(pb, "_ddl_arena", None)
                    if arena is not None and arena.grad_is_arena_view(pb):
                        K.slab_fold(slab, pb.grad, accumulate=True)
                        arena.notify_grad_ready(pb)
                    else:
                        grads[2 * (i - 1) + 1] = K.slab_fold(slab, torch.empty_like(pb, dtype=torch.float32))
                    gated = True
                else:
                    dx = K.conv2d_dgrad(d, crsk, tuple(x_in.shape[2:]), stride, pad)
            if not _WGRAD_FIRST_PLAIN:
                grads[2 * i] = _wgrad(conv.weight, d, x_in, R, S, stride, pad, conv.in_channels)
            d = dx
        return (d, None, *grads)


class FusedMLP(torch.autograd.Function):
    """A whole stack of ``Linear (+ ReLU)`` layers up to 64 wide in ONE forward and ONE backward launch
    (``csrc/mlp.hip``), fp32.  ``FusedMLP.apply(x, relu_last, w0, b0, w1, b1, ...)``; ReLU follows every layer but the
    last (and the last too with ``relu_last``).  Parameter gradients are accumulated straight into the arena when the
    parameters live there (else returned to autograd)."""

    @staticmethod
    def forward(ctx, x, relu_last, *params):
        ws, bs = list(params[0::2]), list(params[1::2])
        xf = x.float().contiguous()
        need = any(ctx.needs_input_grad)
        out, acts = K.mlp_fwd(xf, [w.detach() for w in ws], [None if b is None else b.detach() for b in bs],
                              relu_last=relu_last, save=need)
        if need:
            ctx.save_for_backward(out, acts)
            ctx.params = (ws, bs)
            ctx.relu_last = relu_last
            ctx.x_dtype = x.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        out, acts = ctx.saved_tensors
        ws, bs = ctx.params
        direct, gw, gb = [], [], []
        for p in ws + bs:                                   # arena-backed parameters: accumulate in place
            arena = getattr(p, "_ddl_arena", None) if p is not None else None
            direct.append(p is not None and arena is not None and arena.grad_is_arena_view(p) and p.grad.is_contiguous())
        n = len(ws)
        for i, w in enumerate(ws):
            gw.append(w.grad if direct[i] else (torch.zeros_like(w, dtype=torch.float32) if w.requires_grad else None))
        for i, b in enumerate(bs):
            if b is None:
                gb.append(None)
            else:
                gb.append(b.grad if direct[n + i] else (torch.zeros_like(b, dtype=torch.float32) if b.requires_grad else None))
        dx = K.mlp_bwd(dout.float().contiguous(), out, acts, [w.detach() for w in ws], gw, gb, relu_last=ctx.relu_last,
                       want_dx=ctx.needs_input_grad[0])
        grads = []
        for i in range(n):
            for p, g, d in ((ws[i], gw[i], direct[i]), (bs[i], gb[i], direct[n + i])):
                if p is not None and d:
                    p._ddl_arena.notify_grad_ready(p)       # noqa: SLF001
                    grads.append(None)
                else:
                    grads.append(g)
        return (dx.to(ctx.x_dtype) if dx is not None else None, None, *grads)


class BNAct(torch.autograd.Function):
    """Stand-alone training BatchNorm (+ residual)(+ ReLU) for inputs not produced by our conv (e.g. concat)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, momentum, eps, relu, slabs):
        x = K.nhwc(_bf16(x))
        N, _, H, W = x.shape
        slab = K.bn_stats(x, slab=slabs[0] if slabs else None)
        ctx.slab = slabs[1] if slabs else None
        ctx.bn_params = (gamma, beta)
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
        dx, dres, dgamma, dbeta = _bn_bwd(dout, mask, x, ctx.bn_params[0], ctx.bn_params[1], gamma, vec[0], vec[1],
                                          relu, has_res, ctx.slab)
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


class LinearAct(torch.autograd.Function):
    """``y = dropout(relu(x @ W^T + b))`` with all three GEMMs on the hand-written split-M MFMA kernel
    (``ops.kernels.gemm_tn``; see ``csrc/linear.hip`` for why that kernel and how the operands are laid out), bias /
    ReLU / dropout in the epilogue kernel, ReLU+dropout gate and bias gradient in ONE backward pass, and the weight
    gradient accumulated straight into the arena on the side stream.

    ``w16`` bf16 ``[out_p, in_p]`` and ``w16_t`` bf16 ``[in_p, out_p]`` are the (8-padded) weight and its transpose —
    the arena's shadow pair when the parameter lives there.  No dropout mask is stored: behind a ReLU a kept, live
    element is one with ``y > 0``; dropout WITHOUT ReLU keeps negative values too, there the gate is ``y != 0``.
    Reference: every model of the reference ends in ``nn.Linear``
    (``/root/reference/src/pytorch/CNN/model.py:183``, ``MLP/model.py:51-58``, ``LSTM/model.py:94``)."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16, w16_t, bias_p, relu, drop_p):
        B, fin = x.shape
        out_f = weight.shape[0]
        out_p, in_p = w16.shape
        Bp = (B + 7) // 8 * 8
        if x.dtype not in (BF16, torch.float32) or x.stride(1) != 1:
            x = x.to(BF16).contiguous()
        if fin != in_p:                                    # feature count not a multiple of 8 (LeNet's 84): zero columns
            x = torch.nn.functional.pad(x, (0, in_p - fin))
        xt = K.transpose_pad(x, Bp)                        # [in_p, Bp]: the reduction index leads
        acc = K.zero_accumulator(Bp, out_p, x.device)      # persistent, handed back zeroed by the epilogue
        try:
            K.gemm_tn(xt, w16_t, acc)
            y = K.linear_epilogue(acc, bias_p, Bp, relu=relu, drop_p=drop_p, clear=True)
        except Exception:
            K.drop_accumulators()
            raise
        need_w = ctx.needs_input_grad[1]
        xs = None
        if need_w:                                         # wgrad operand: bf16 [Bp, in_p], batch rows zero-padded
            xs = x if (x.dtype == BF16 and x.is_contiguous() and Bp == B) else \
                torch.nn.functional.pad(x.to(BF16), (0, 0, 0, Bp - B)).contiguous()
        gated = relu or drop_p > 0.0
        ctx.save_for_backward(xs, y if gated else None, w16)
        ctx.meta = (B, fin, out_f, gated, 1.0 / (1.0 - drop_p) if drop_p > 0.0 else 1.0)
        ctx.gate_nonzero = bool(drop_p > 0.0 and not relu)
        ctx.weight, ctx.bias = weight, bias
        return y if (Bp == B and out_p == out_f) else y[:B, :out_f]

    @staticmethod
    def backward(ctx, dy):
        xs, y, w16 = ctx.saved_tensors
        B, fin, out_f, gated, scale = ctx.meta
        out_p, in_p = w16.shape
        weight, bias = ctx.weight, ctx.bias
        dy = _bf16(dy)
        Bp = (B + 7) // 8 * 8
        if Bp != B or out_p != out_f:
            dy = torch.nn.functional.pad(dy, (0, out_p - out_f, 0, Bp - B))
        dy = dy.contiguous()
        # gate (ReLU / dropout) + bias gradient in one pass; the bias gradient goes straight into the arena when it can
        db = None
        has_bias = bias is not None and ctx.needs_input_grad[2]
        arena_b = getattr(bias, "_ddl_arena", None) if has_bias else None
        direct_b = (arena_b is not None and out_p == out_f and arena_b.grad_is_arena_view(bias) and bias.grad.is_contiguous())
        if gated or has_bias or scale != 1.0:
            g, dbv = K.linear_bwd_gate(dy, y if gated else None, scale, grad_out=bias.grad if direct_b else None,
                                       want_bias=has_bias, want_g=gated or scale != 1.0,
                                       gate_nonzero=ctx.gate_nonzero)
            if g is None:
                g = dy
            if has_bias:
                if direct_b:
                    arena_b.notify_grad_ready(bias)
                else:
                    db = dbv[:out_f]
        else:
            g = dy
        dx = None
        if ctx.needs_input_grad[0]:
            gt = K.transpose_pad(g, Bp)                    # [out_p, Bp]
            acc = K.zero_accumulator(Bp, in_p, g.device)
            try:
                K.gemm_tn(gt, w16, acc)
                dx = K.cast_rows(acc, B, in_p, clear=True)
            except Exception:
                K.drop_accumulators()
                raise
            if in_p != fin:
                dx = dx[:, :fin]
        dw = None
        if ctx.needs_input_grad[1]:
            arena = getattr(weight, "_ddl_arena", None)
            if (arena is not None and out_p == out_f and in_p == fin and arena.grad_is_arena_view(weight)
                    and weight.grad.is_contiguous()):
                with streams.fork(g, xs):                  # side stream: overlaps the rest of backward
                    K.gemm_tn(g, xs, weight.grad)
                arena.notify_grad_ready(weight)
            else:
                acc = torch.zeros((out_p, in_p), dtype=torch.float32, device=g.device)
                K.gemm_tn(g, xs, acc)
                dw = acc[:out_f, :fin]
        return dx, dw, db, None, None, None, None, None


class Conv1dK1Relu(torch.autograd.Function):
    """``relu(Conv1d(kernel_size=1)(x))`` on raw fp32 input ``[B, C, L]`` -> bf16 ``[B, O, L]`` (``csrc/linear.hip``):
    the front of the reference's LSTM workload (``/root/reference/src/pytorch/LSTM/model.py:70-71``).  ``weight`` is the
    Conv1d parameter ``[O, C, 1]``; its gradient and the bias gradient are accumulated straight into the arena when
    the parameters live there.  The input is data: no input gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        w2 = weight.detach().reshape(weight.shape[0], weight.shape[1])
        y = K.conv1d_k1_relu_fwd(x, w2, None if bias is None else bias.detach())
        ctx.save_for_backward(x, y)
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        weight, bias = ctx.params
        dy = _bf16(dy).contiguous()
        arena = getattr(weight, "_ddl_arena", None)
        direct = (arena is not None and arena.grad_is_arena_view(weight) and weight.grad.is_contiguous()
                  and (bias is None or (getattr(bias, "_ddl_arena", None) is arena and arena.grad_is_arena_view(bias))))
        if direct:
            K.conv1d_k1_relu_bwd(dy, y, x, weight.grad, None if bias is None else bias.grad)
            arena.notify_grad_ready(weight)
            if bias is not None:
                arena.notify_grad_ready(bias)
            return None, None, None
        dw = torch.zeros(weight.shape, dtype=torch.float32, device=dy.device)
        db = torch.zeros(bias.shape, dtype=torch.float32, device=dy.device) if bias is not None else None
        K.conv1d_k1_relu_bwd(dy, y, x, dw, db)
        return None, dw, db


class L1Loss(torch.autograd.Function):
    """Mean absolute error with the gradient produced in the forward launch (``csrc/linear.hip``): the reference's
    LSTM criterion, ``/root/reference/src/pytorch/LSTM/main.py:163``."""

    @staticmethod
    def forward(ctx, pred, target):
        out, grad = K.l1_loss(pred, target, want_grad=ctx.needs_input_grad[0])
        ctx.save_for_backward(grad)
        ctx.pred_dtype = pred.dtype
        return out[0]

    @staticmethod
    def backward(ctx, gloss):
        (grad,) = ctx.saved_tensors
        return (grad * gloss).to(ctx.pred_dtype), None


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
