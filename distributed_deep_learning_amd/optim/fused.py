"""Arena optimizers: SGD-momentum, LARS and Adam as ONE fused launch over the flat parameter arena.

GPU path  : ``csrc/optim.hip`` — per 1024-element chunk the kernel looks up its tensor for weight decay / LARS
            trust ratio, updates fp32 master + state, rewrites the bf16 shadow the conv kernels read, and clears
            the gradient.  Hyper-parameters live in a device array so LR schedules need no host sync.
CPU path  : the same math with flat torch ops (gloo plumbing config, unit tests).

Both support a chunk sub-range ``[lo, hi)`` so the PS-style mode (reduce-scatter -> update own shard ->
all-gather) reuses them unchanged.

Reference: ``SGD(lr=0.01, momentum=0.9)`` at ``/root/reference/src/pytorch/CNN/main.py:160``, ``Adam()`` at
``/root/reference/src/pytorch/LSTM/main.py:164``; LARS is a north-star addition (BASELINE.json).
"""
from __future__ import annotations

import math
from typing import Callable, Iterable, List, Optional

import torch

from ..ops import _lib
from ..ops import state as _state
from ..ops import streams as _streams
from ..parallel.arena import CHUNK, ParamArena


def default_no_decay(name: str, p: torch.Tensor) -> bool:
    """Biases and normalisation parameters (1-D tensors) are excluded from weight decay and LARS adaptation."""
    return p.dim() <= 1


class _ArenaOptimizer(torch.optim.Optimizer):
    """Common machinery: arena binding, per-segment tables, range handling, state (de)serialisation."""

    _state_names: List[str] = []

    def __init__(self, arena_or_params, defaults: dict, no_decay: Optional[Callable] = default_no_decay,
                 bucket_bytes: int = 32 << 20):
        if isinstance(arena_or_params, ParamArena):
            arena = arena_or_params
        else:
            named = list(arena_or_params)
            if named and not isinstance(named[0], tuple):
                named = [(f"param{i}", p) for i, p in enumerate(named)]
            existing = {getattr(p, "_ddl_arena", None) for _, p in named}
            if len(existing) == 1 and None not in existing:
                arena = existing.pop()
            else:
                arena = ParamArena(named, bucket_bytes=bucket_bytes)
        self.arena = arena
        super().__init__(arena.params, defaults)
        dev = arena.device
        wd = float(defaults.get("weight_decay", 0.0))
        seg_wd = torch.zeros(len(arena.params), dtype=torch.float32)
        seg_adapt = torch.zeros(len(arena.params), dtype=torch.int32)
        for i, (n, p) in enumerate(zip(arena.names, arena.params)):
            skip = no_decay(n, p) if no_decay is not None else False
            seg_wd[i] = 0.0 if skip else wd
            seg_adapt[i] = 0 if skip else 1
        self.seg_wd = seg_wd.to(dev)
        self.seg_adapt = seg_adapt.to(dev)
        self._seg_wd_host = seg_wd
        self._seg_adapt_host = seg_adapt
        self.flat_state = {k: torch.zeros(arena.total, dtype=torch.float32, device=dev) for k in self._state_names}
        self._shard: Optional[List] = None      # PS / ZeRO-1: [(lo, hi, compact offset)] once shard_state() ran
        self._gather_state: Optional[Callable] = None
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        self._hyper_host = None
        self.grad_scale = 1.0
        self.step_count = 0
        self._wd_vec = None       # CPU path: per-element weight decay

    # ---- helpers
    @property
    def lr(self) -> float:
        return float(self.param_groups[0]["lr"])

    def _set_hyper(self, vals: List[float]) -> None:
        vals = [float(v) for v in vals] + [0.0] * (8 - len(vals))
        if vals != self._hyper_host:
            self.hyper.copy_(torch.tensor(vals, dtype=torch.float32), non_blocking=True)
            self._hyper_host = vals

    def _chunks(self, lo: Optional[int], hi: Optional[int]):
        lo = 0 if lo is None else lo
        hi = self.arena.total if hi is None else hi
        if lo % CHUNK or hi % CHUNK:
            raise ValueError("optimizer ranges must be chunk aligned")
        return lo // CHUNK, hi // CHUNK

    def _elem_vec(self, seg_values: torch.Tensor) -> torch.Tensor:
        """Expand a per-segment host table to a per-element vector on the arena device (CPU path only)."""
        a = self.arena
        v = torch.zeros(a.total, dtype=torch.float32)
        for s in a.segments:
            v[s.offset:s.offset + s.numel] = float(seg_values[s.index])
        return v.to(a.device)

    # ---- sharded state (PS / ZeRO-1 mode): buffers cover only the element ranges this rank updates
    def shard_state(self, ranges, gather: Optional[Callable] = None) -> None:
        """Keep optimizer state only for ``ranges`` (the rank's shard of every bucket, chunk aligned): each state
        tensor shrinks from ``arena.total`` to ``sum(hi - lo)`` elements, the owned ranges back to back.  ``gather``:
        collective callable (compact tensor -> full arena-layout tensor) used by :meth:`state_dict`."""
        shard, off = [], 0
        for lo, hi in ranges:
            if lo % CHUNK or hi % CHUNK:
                raise ValueError("optimizer ranges must be chunk aligned")
            shard.append((lo, hi, off))
            off += hi - lo
        old, old_shard = self.flat_state, self._shard
        new = {k: torch.zeros(off, dtype=torch.float32, device=self.arena.device) for k in self._state_names}
        for k in self._state_names:
            for lo, hi, coff in shard:
                if old_shard is None:
                    new[k][coff:coff + hi - lo].copy_(old[k][lo:hi])
        self.flat_state, self._shard, self._gather_state = new, shard, gather

    def _state_base(self, name: str, lo: int) -> int:
        """Device address the kernels index with ABSOLUTE arena offsets so that element ``lo`` lands on this range's
        slot of the (possibly compact) state buffer; only addresses inside the buffer are ever dereferenced."""
        buf = self.flat_state[name]
        if self._shard is None:
            return buf.data_ptr()
        for slo, shi, coff in self._shard:
            if slo <= lo < shi:
                return buf.data_ptr() + 4 * (coff - slo)
        raise ValueError(f"optimizer state is sharded and holds nothing for element {lo}")

    def _state_slice(self, name: str, lo: int, hi: int) -> torch.Tensor:
        buf = self.flat_state[name]
        if self._shard is None:
            return buf[lo:hi]
        for slo, shi, coff in self._shard:
            if slo <= lo and hi <= shi:
                return buf[coff + lo - slo:coff + hi - slo]
        raise ValueError(f"optimizer state is sharded and holds nothing for [{lo}, {hi})")

    def state_bytes(self) -> int:
        return sum(v.numel() * v.element_size() for v in self.flat_state.values())

    def zero_grad(self, set_to_none: bool = False) -> None:  # noqa: D401 - torch signature
        """Gradients live in the arena; clearing is one memset (and the fused step already does it)."""
        self.arena.zero_grad()

    def _use_kernel(self) -> bool:
        return self.arena.device.type == "cuda"

    def step(self, closure=None, ranges=None, zero_grad: bool = True):
        """``ranges``: optional list of element ranges ``(lo, hi)`` (PS mode); default = whole arena."""
        loss = closure() if closure is not None else None
        with _lib.device_of(self.arena.device):     # per-stage optimizers (model parallel): launch on the arena's GPU
            _streams.join()             # side-stream filter gradients must have landed in the arena
            self.host_prepare()
            self.launch(ranges, zero_grad)
            self.host_finish()
        return loss

    # The step in three parts, so a captured HIP graph (utils/graph.py) can hold ONLY the device part: the host
    # parts run around every replay (hyper-parameters reach the kernels through the device-resident ``hyper`` array).
    def host_prepare(self) -> None:
        self.step_count += 1
        self._prepare()

    def launch(self, ranges=None, zero_grad: bool = True) -> None:
        self._before_ranges(ranges)
        for lo, hi in (ranges if ranges is not None else [(None, None)]):
            clo, chi = self._chunks(lo, hi)
            if chi > clo:
                self._update(clo, chi, zero_grad)
        if self.arena.shadow is not None and self._use_kernel():
            self.arena.refresh_transposes()

    def host_finish(self) -> None:
        _state.bump()
        if self.arena.shadow is not None and self._use_kernel():
            self.arena.mark_shadow_current()

    # ---- subclass API
    def _prepare(self) -> None:
        raise NotImplementedError

    def _before_ranges(self, ranges) -> None:
        pass

    def _update(self, clo: int, chi: int, zero_grad: bool) -> None:
        raise NotImplementedError

    # ---- (de)serialisation: flat state + scalars
    def state_dict(self):
        """Full (un-sharded) flat state in arena layout.  With sharded state this is a COLLECTIVE: every rank calls it
        and every rank gets the complete state (the shards are all-gathered through the reducer)."""
        if self._shard is not None:
            if self._gather_state is None:
                raise RuntimeError("sharded optimizer state needs the reducer's gather to be serialised")
            flat = {k: self._gather_state(v.detach()).cpu() for k, v in self.flat_state.items()}
        else:
            flat = {k: v.detach().cpu().clone() for k, v in self.flat_state.items()}
        return {
            "kind": type(self).__name__,
            "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
            "flat_state": flat,
            "step_count": self.step_count,
            "layout": [(s.name, s.offset, s.numel) for s in self.arena.segments],
        }

    def load_state_dict(self, sd):
        if sd.get("kind") != type(self).__name__:
            raise ValueError(f"optimizer kind mismatch: {sd.get('kind')} vs {type(self).__name__}")
        layout = [(s.name, s.offset, s.numel) for s in self.arena.segments]
        saved = [tuple(x) for x in sd["layout"]]
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            g.update(sg)
        self.step_count = int(sd["step_count"])
        full = {}
        if saved == layout:
            full = dict(sd["flat_state"])
        else:  # different bucket plan / world size: remap tensor by tensor
            by_name = {n: (o, c) for n, o, c in saved}
            for k, v in sd["flat_state"].items():
                full[k] = torch.zeros(self.arena.total, dtype=torch.float32)
            for s in self.arena.segments:
                if s.name not in by_name:
                    raise KeyError(f"optimizer state for {s.name} missing from checkpoint")
                o, c = by_name[s.name]
                if c != s.numel:
                    raise ValueError(f"optimizer state size mismatch for {s.name}")
                for k, v in sd["flat_state"].items():
                    full[k][s.offset:s.offset + c].copy_(v[o:o + c])
        for k, v in full.items():
            if self._shard is None:
                self.flat_state[k].copy_(v)
            else:                                   # sharded: every rank keeps its own ranges of the full state
                for lo, hi, coff in self._shard:
                    self.flat_state[k][coff:coff + hi - lo].copy_(v[lo:hi])
        self._hyper_host = None


class FusedSGD(_ArenaOptimizer):
    """SGD with momentum (PyTorch semantics) and optional LARS layer-wise trust ratios.

    ``lars=True``: for every tensor not excluded by ``no_decay``
    ``trust = eta * ||w|| / (||g|| + wd * ||w|| + eps)`` scales ``g + wd*w`` before the momentum update.
    ``norm_reduce``: callable ``f(norms[nseg,2])`` that sums the squared norms over ranks — used by the PS mode,
    where each rank only sees its shard of every tensor.
    """

    _state_names = ["momentum"]

    def __init__(self, arena_or_params, lr: float = 0.01, momentum: float = 0.9, weight_decay: float = 0.0,
                 dampening: float = 0.0, nesterov: bool = False, lars: bool = False, lars_eta: float = 0.001,
                 lars_eps: float = 1e-9, no_decay: Optional[Callable] = default_no_decay, **kw):
        defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=dampening,
                        nesterov=nesterov, lars=lars, lars_eta=lars_eta, lars_eps=lars_eps)
        super().__init__(arena_or_params, defaults, no_decay=no_decay, **kw)
        self.norms = torch.zeros((len(self.arena.params), 2), dtype=torch.float32, device=self.arena.device)
        self.norm_reduce: Optional[Callable] = None
        self._trust_elem = None

    def _prepare(self) -> None:
        g = self.param_groups[0]
        self._set_hyper([g["lr"], g["momentum"], g["dampening"], 1.0 if g["nesterov"] else 0.0, self.grad_scale,
                         g["lars_eta"], g["lars_eps"]])
        self._norms_ready = False

    def _before_ranges(self, ranges) -> None:
        if self.param_groups[0]["lars"]:
            self.compute_norms(ranges)

    def compute_norms(self, ranges=None) -> None:
        """Per-tensor sum w^2 / sum g^2 over the given element ranges (LARS pre-pass)."""
        a = self.arena
        rngs = ranges if ranges is not None else [(0, a.total)]
        if self._use_kernel():
            lib = _lib.require()
            total = None
            for lo, hi in rngs:
                clo, chi = self._chunks(lo, hi)
                part = self.norms if total is None else torch.empty_like(self.norms)
                _lib.check(lib.ddl_flat_sumsq(a.master.data_ptr(), a.grad.data_ptr(), a.chunk_seg.data_ptr(),
                                              part.data_ptr(), len(a.params), clo, chi, _lib.stream_ptr()),
                           "flat_sumsq")
                if total is None:
                    total = part
                else:
                    total += part
        else:
            self.norms.zero_()
            for s in a.segments:
                for lo, hi in rngs:
                    l, h = max(lo, s.offset), min(hi, s.offset + s.numel)
                    if h > l:
                        self.norms[s.index, 0] += a.master[l:h].double().pow(2).sum().float()
                        self.norms[s.index, 1] += a.grad[l:h].double().pow(2).sum().float()
        if self.norm_reduce is not None:
            self.norm_reduce(self.norms)
        self._norms_ready = True

    def _update(self, clo: int, chi: int, zero_grad: bool) -> None:
        g = self.param_groups[0]
        a = self.arena
        use_mom = g["momentum"] != 0.0
        if self._use_kernel():
            lib = _lib.require()
            _lib.check(lib.ddl_flat_sgd(a.master.data_ptr(), a.grad.data_ptr(),
                                        self._state_base("momentum", clo * CHUNK) if use_mom else None,
                                        a.shadow.data_ptr() if a.shadow is not None else None,
                                        a.chunk_seg.data_ptr(), self.seg_wd.data_ptr(), self.seg_adapt.data_ptr(),
                                        self.norms.data_ptr() if g["lars"] else None, self.hyper.data_ptr(),
                                        clo, chi, int(zero_grad), _lib.stream_ptr()), "flat_sgd")
            return
        lo, hi = clo * CHUNK, chi * CHUNK
        w, gr = a.master[lo:hi], a.grad[lo:hi]
        if self._wd_vec is None:
            self._wd_vec = self._elem_vec(self._seg_wd_host)
        gp = gr * self.grad_scale + self._wd_vec[lo:hi] * w
        if g["lars"]:
            wn = self.norms[:, 0].sqrt()
            gn = self.norms[:, 1].sqrt() * abs(self.grad_scale)
            trust = g["lars_eta"] * wn / (gn + self.seg_wd * wn + g["lars_eps"])
            trust = torch.where((wn > 0) & (gn > 0) & (self.seg_adapt > 0), trust, torch.ones_like(trust))
            gp = gp * self._elem_vec(trust.cpu())[lo:hi]
        if use_mom:
            m = self._state_slice("momentum", lo, hi)
            m.mul_(g["momentum"]).add_(gp, alpha=1.0 - g["dampening"])
            stepv = gp + g["momentum"] * m if g["nesterov"] else m
        else:
            stepv = gp
        w.add_(stepv, alpha=-g["lr"])
        if a.shadow is not None:
            a.shadow[lo:hi].copy_(w)
        if zero_grad:
            gr.zero_()


class LARS(FusedSGD):
    """Layer-wise Adaptive Rate Scaling (You et al., 2017) for large-batch ResNet-50 training."""

    def __init__(self, arena_or_params, lr: float = 0.1, momentum: float = 0.9, weight_decay: float = 5e-5,
                 eta: float = 0.001, eps: float = 1e-9, **kw):
        super().__init__(arena_or_params, lr=lr, momentum=momentum, weight_decay=weight_decay, lars=True,
                         lars_eta=eta, lars_eps=eps, **kw)


class FusedAdam(_ArenaOptimizer):
    _state_names = ["exp_avg", "exp_avg_sq"]

    def __init__(self, arena_or_params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, no_decay: Optional[Callable] = default_no_decay, **kw):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(arena_or_params, defaults, no_decay=no_decay, **kw)

    def _prepare(self) -> None:
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        t = self.step_count
        self._set_hyper([g["lr"], b1, b2, g["eps"], self.grad_scale, 1.0 - b1 ** t, 1.0 - b2 ** t])

    def _update(self, clo: int, chi: int, zero_grad: bool) -> None:
        g = self.param_groups[0]
        a = self.arena
        if self._use_kernel():
            lib = _lib.require()
            _lib.check(lib.ddl_flat_adam(a.master.data_ptr(), a.grad.data_ptr(),
                                         self._state_base("exp_avg", clo * CHUNK),
                                         self._state_base("exp_avg_sq", clo * CHUNK),
                                         a.shadow.data_ptr() if a.shadow is not None else None,
                                         a.chunk_seg.data_ptr(), self.seg_wd.data_ptr(), self.hyper.data_ptr(), clo,
                                         chi, int(zero_grad), _lib.stream_ptr()), "flat_adam")
            return
        lo, hi = clo * CHUNK, chi * CHUNK
        b1, b2 = g["betas"]
        t = self.step_count
        w, gr = a.master[lo:hi], a.grad[lo:hi]
        if self._wd_vec is None:
            self._wd_vec = self._elem_vec(self._seg_wd_host)
        gp = gr * self.grad_scale + self._wd_vec[lo:hi] * w
        m, v = self._state_slice("exp_avg", lo, hi), self._state_slice("exp_avg_sq", lo, hi)
        m.mul_(b1).add_(gp, alpha=1 - b1)
        v.mul_(b2).addcmul_(gp, gp, value=1 - b2)
        denom = (v / (1 - b2 ** t)).sqrt_().add_(g["eps"])
        w.addcdiv_(m / (1 - b1 ** t), denom, value=-g["lr"])
        if a.shadow is not None:
            a.shadow[lo:hi].copy_(w)
        if zero_grad:
            gr.zero_()
