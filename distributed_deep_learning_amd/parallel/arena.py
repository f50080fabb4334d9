"""Flat parameter arena: the memory layout everything distributed is built on.

All parameters of a model (per device) live in ONE contiguous fp32 master buffer, with three siblings of the
same layout: gradients (fp32), optimizer state, and — on the GPU — a bf16 shadow the conv kernels read.
``Parameter.data`` / ``Parameter.grad`` become strided views into the arena (memory format preserved, so
``state_dict`` is unchanged).  Consequences, all chosen for MI355X (288 GB HBM, 7 xGMI links per GPU):

* gradient buckets are contiguous arena ranges -> RCCL all-reduce / reduce-scatter / all-gather run in place on
  them, no pack / unpack / flatten copies, few large collectives;
* the "multi-tensor" optimizer is a single kernel launch over the range (csrc/optim.hip) which also rewrites the
  bf16 shadow and clears the gradient;
* the PS-style (ZeRO-1) mode shards each bucket evenly: buckets are padded to ``shard_multiple`` chunks.

Layout: tensors start on ``CHUNK``-element boundaries; consecutive tensors are grouped into buckets of about
``bucket_bytes`` (the LAST bucket — first to become ready in backward — is kept smaller so communication starts
early); each bucket is padded to a multiple of ``CHUNK * shard_multiple``.

Reference: replaces the per-parameter ``all_reduce`` loop of ``/root/reference/src/pytorch/CNN/main.py:136-139``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence

import torch

from ..ops import _lib

CHUNK = _lib.chunk_elems()


@dataclass
class Segment:
    """One parameter tensor inside the arena."""
    index: int
    name: str
    offset: int          # element offset of the tensor
    numel: int
    padded: int          # elements reserved (multiple of CHUNK)
    bucket: int


@dataclass
class Bucket:
    index: int
    lo: int              # element range [lo, hi) — hi includes the shard padding
    hi: int
    segments: List[int]


def _ceil_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def plan_buckets(sizes: Sequence[int], bucket_elems: int, first_bucket_elems: Optional[int] = None) -> List[List[int]]:
    """Group tensor indices (registration order) into buckets.  Buckets are filled from the END of the list,
    because backward produces gradients in reverse registration order; the bucket that becomes ready first is
    capped at ``first_bucket_elems`` so its collective starts early.  Returns buckets in registration order."""
    if first_bucket_elems is None:
        first_bucket_elems = max(bucket_elems // 4, 1)
    buckets: List[List[int]] = []
    cur: List[int] = []
    cur_elems = 0
    cap = first_bucket_elems
    for i in range(len(sizes) - 1, -1, -1):
        cur.append(i)
        cur_elems += sizes[i]
        if cur_elems >= cap:
            buckets.append(list(reversed(cur)))
            cur, cur_elems, cap = [], 0, bucket_elems
    if cur:
        buckets.append(list(reversed(cur)))
    buckets.reverse()
    return buckets


class ParamArena:
    def __init__(self, named_params: Iterable, bucket_bytes: int = 32 << 20, shard_multiple: int = 1,
                 shadow: Optional[bool] = None, first_bucket_bytes: Optional[int] = None):
        named = [(n, p) for n, p in named_params if p.requires_grad]
        if not named:
            raise ValueError("ParamArena: no trainable parameters")
        dev = named[0][1].device
        for n, p in named:
            if p.device != dev:
                raise ValueError(f"ParamArena: parameter {n} is on {p.device}, expected {dev} (one arena per device)")
            if p.dtype != torch.float32:
                raise TypeError(f"ParamArena: parameter {n} must be fp32 (master weights), got {p.dtype}")
        self.device = dev
        self.shard_multiple = int(shard_multiple)
        self.params = [p for _, p in named]
        self.names = [n for n, _ in named]
        padded = [_ceil_to(p.numel(), CHUNK) for p in self.params]
        groups = plan_buckets(padded, max(bucket_bytes // 4, CHUNK),
                              None if first_bucket_bytes is None else max(first_bucket_bytes // 4, CHUNK))
        self.segments: List[Segment] = [None] * len(self.params)  # type: ignore[list-item]
        self.buckets: List[Bucket] = []
        off = 0
        for b, idxs in enumerate(groups):
            lo = off
            for i in idxs:
                self.segments[i] = Segment(i, self.names[i], off, self.params[i].numel(), padded[i], b)
                off += padded[i]
            off = lo + _ceil_to(off - lo, CHUNK * self.shard_multiple)
            self.buckets.append(Bucket(b, lo, off, list(idxs)))
        self.total = off
        self.nchunks = self.total // CHUNK

        self.master = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        use_shadow = dev.type == "cuda" if shadow is None else shadow
        self.shadow = torch.zeros(self.total, dtype=torch.bfloat16, device=dev) if use_shadow else None

        chunk_seg = torch.full((self.nchunks,), -1, dtype=torch.int32)
        for s in self.segments:
            chunk_seg[s.offset // CHUNK:(s.offset + s.padded) // CHUNK] = s.index
        self.chunk_seg = chunk_seg.to(dev)

        with torch.no_grad():
            for p, s in zip(self.params, self.segments):
                if not _dense(p):
                    p.data = p.data.contiguous()
                mview = torch.as_strided(self.master, p.shape, p.stride(), s.offset)
                mview.copy_(p.data)
                p.data = mview
                p.grad = torch.as_strided(self.grad, p.shape, p.stride(), s.offset)
                p._ddl_arena = self                                   # noqa: SLF001
                p._ddl_segment = s.index                              # noqa: SLF001
                if self.shadow is not None:
                    p._ddl_shadow = torch.as_strided(self.shadow, p.shape, p.stride(), s.offset)  # noqa: SLF001
        self._shadow_versions = [-1] * len(self.params)
        self.on_grad_ready = None          # set by the DP reducer: called with a segment index
        self.shadow_t = None               # bf16 [C][R][S][K] repacks of every conv filter (dgrad operand)
        self._repack = None
        if self.shadow is not None and dev.type == "cuda":
            self._build_repack_table()
        self.sync_shadow()

    # ------------------------------------------------------------------ conv filter repacks (one launch per step)
    def _build_repack_table(self) -> None:
        """Descriptor table for ``csrc/eltwise.hip::filter_transpose_batched_kernel``: every 4-D parameter stored
        channels-last (physically KRSC) with C % 8 == 0 and K % 8 == 0 gets a CRSK twin at the same arena offset, and
        every contiguous 2-D (Linear) weight ``[out, in]`` with both sides % 8 == 0 an ``[in, out]`` twin (a 1x1 filter)."""
        import numpy as np
        dt = np.dtype([("in_off", "<i8"), ("out_off", "<i8"), ("K", "<i4"), ("RS", "<i4"), ("C", "<i4"),
                       ("tile_begin", "<i4"), ("tiles_c", "<i4"), ("tiles_k", "<i4")])
        def geom(p):
            if p.dim() == 4 and not (p.shape[1] % 8 or p.shape[0] % 8) and p.permute(0, 2, 3, 1).is_contiguous():
                return p.shape
            if p.dim() == 2 and not (p.shape[1] % 8 or p.shape[0] % 8) and p.is_contiguous():
                return (p.shape[0], p.shape[1], 1, 1)
            return None

        rows, tile = [], 0
        for p, s in zip(self.params, self.segments):
            if geom(p) is None:
                continue
            K_, C_, R_, S_ = geom(p)
            tc, tk = (C_ + 31) // 32, (K_ + 31) // 32
            rows.append((s.offset, s.offset, K_, R_ * S_, C_, tile, tc, tk))
            tile += R_ * S_ * tc * tk
        if not rows:
            return
        assert dt.itemsize == _lib.require().ddl_repack_desc_bytes()
        arr = np.array(rows, dtype=dt)
        self.shadow_t = torch.zeros(self.total, dtype=torch.bfloat16, device=self.device)
        descs = torch.from_numpy(arr.view(np.uint8).copy()).to(self.device)
        self._repack = (descs, len(rows), tile)
        for p, s in zip(self.params, self.segments):
            if geom(p) is not None:
                K_, C_, R_, S_ = geom(p)
                twin = self.shadow_t[s.offset:s.offset + s.numel]
                p._ddl_shadow_t = twin.view(C_, R_, S_, K_) if p.dim() == 4 else twin.view(C_, K_)   # noqa: SLF001

    def refresh_transposes(self) -> None:
        if self._repack is not None:
            from ..ops import kernels as K
            descs, n, tiles = self._repack
            K.filter_transpose_batched(self.shadow, self.shadow_t, descs, n, tiles)

    def notify_grad_ready(self, p) -> None:
        """Called by kernels that accumulate straight into the arena gradient and hand ``None`` to autograd.  (Autograd
        still runs the parameter's post-accumulate hook for that undefined gradient, later; the reducer counts the
        first signal per tensor and pass.)"""
        if self.on_grad_ready is not None:
            self.on_grad_ready(p._ddl_segment)                        # noqa: SLF001

    def grad_is_arena_view(self, p) -> bool:
        g = p.grad
        return g is not None and g.data_ptr() == self.grad.data_ptr() + 4 * self.segments[p._ddl_segment].offset  # noqa: SLF001

    # ------------------------------------------------------------------ views
    def grad_range(self, lo: int, hi: int) -> torch.Tensor:
        return self.grad[lo:hi]

    def bucket_grad(self, b: int) -> torch.Tensor:
        bk = self.buckets[b]
        return self.grad[bk.lo:bk.hi]

    def bucket_master(self, b: int) -> torch.Tensor:
        bk = self.buckets[b]
        return self.master[bk.lo:bk.hi]

    def shard_range(self, b: int, rank: int, world: int):
        """Element range of bucket ``b`` owned by ``rank`` in the PS / ZeRO-1 mode."""
        bk = self.buckets[b]
        n = (bk.hi - bk.lo) // world
        return bk.lo + rank * n, bk.lo + (rank + 1) * n

    # ------------------------------------------------------------------ maintenance
    def zero_grad(self) -> None:
        from ..ops import streams
        streams.join()                 # no side-stream wgrad may still be accumulating
        self.grad.zero_()

    def reattach_grads(self) -> None:
        """Re-point ``p.grad`` at the arena (after someone called ``zero_grad(set_to_none=True)``)."""
        for p, s in zip(self.params, self.segments):
            g = p.grad
            if g is None or g.data_ptr() != self.grad.data_ptr() + 4 * s.offset:
                view = torch.as_strided(self.grad, p.shape, p.stride(), s.offset)
                if g is not None:
                    view.copy_(g)
                p.grad = view

    def sync_shadow(self) -> None:
        """Refresh the bf16 shadow from the fp32 master (after init, broadcast, load_state_dict, all-gather)."""
        if self.shadow is not None:
            with _lib.device_of(self.device):
                if self.device.type == "cuda" and _lib.available():
                    from ..ops import kernels as K
                    K.cast_f32_bf16(self.master, self.shadow)
                else:
                    self.shadow.copy_(self.master)
                self.refresh_transposes()
        self.mark_shadow_current()

    def mark_shadow_current(self) -> None:
        for i, p in enumerate(self.params):
            self._shadow_versions[i] = p._version
            p._ddl_shadow_version = p._version                        # noqa: SLF001

    def check_layout(self) -> None:
        """Debug aid: every parameter / gradient must still alias the arena."""
        for p, s in zip(self.params, self.segments):
            if p.data_ptr() != self.master.data_ptr() + 4 * s.offset:
                raise RuntimeError(f"parameter {s.name} no longer aliases the arena (was .data reassigned?)")

    def describe(self) -> str:
        mb = self.total * 4 / 2**20
        return (f"ParamArena({len(self.params)} tensors, {self.total} elems = {mb:.1f} MiB fp32, "
                f"{len(self.buckets)} buckets: " + ", ".join(f"{(b.hi - b.lo) * 4 / 2**20:.1f}" for b in self.buckets) + " MiB)")


def _dense(p: torch.Tensor) -> bool:
    """True when the tensor's strides describe a dense, non-overlapping layout (contiguous in some dim order)."""
    if p.numel() == 0:
        return True
    dims = sorted(((st, sz) for st, sz in zip(p.stride(), p.shape) if sz > 1))
    expect = 1
    for st, sz in dims:
        if st != expect:
            return False
        expect *= sz
    return True
