"""Parameter-server-style synchronisation (the idiom of the reference's empty MXNet / TensorFlow / Paddle stubs,
``/root/reference/src/{mxnet,tensorflow,paddle}/**``, SURVEY §2.3 "PS-style sync") — a ZeRO-1 sharded data-parallel
step.  There is no server process: every rank IS the server of its shard.  Per bucket of the flat arena

    push    reduce_scatter(grad bucket)          each rank holds the averaged gradient of ITS shard only
    update  fused SGD / LARS / Adam on the shard  fp32 master + optimizer state exist for 1/world of the parameters
                                                  (``optimizer.shard_state``); LARS norms of tensors cut by shard
                                                  borders are summed over ranks
    pull    all_gather(updated shards)            GPU: the bf16 SHADOW travels (half the bytes of the fp32 master);
                                                  the few 1-D parameters that kernels read in fp32 (BN affine, biases)
                                                  follow in one small fp32 all-reduce; CPU (no shadow): fp32 master

which on the fully connected xGMI mesh moves S/N bytes to each of the 7 peers at once instead of walking a ring.

What each rank holds after a step on the GPU: exact fp32 masters for its own shards and for every 1-D tensor, the
bf16-rounded value (``float(shadow)``) for the rest — every replica computes with identical numbers (filters are
consumed in bf16 anyway).  ``prepare_checkpoint()`` all-gathers the fp32 master shards and the optimizer-state shards
when a checkpoint is written, so checkpoints are exact, complete and independent of the world size that wrote them.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as td

from ..ops._lib import device_of
from .ddp import DistributedDataParallel


class ShardedDataParallel(DistributedDataParallel):
    """``DistributedDataParallel`` on the reduce-scatter -> sharded step -> all-gather schedule (ZeRO-1)."""

    def __init__(self, module, *args, **kw):
        if len(args) > 1:                                   # (bucket_mb, mode, ...) given positionally
            args = (args[0], "ps") + tuple(args[2:])
        else:
            kw["mode"] = "ps"
        super().__init__(module, *args, **kw)
        self._small_idx: Optional[torch.Tensor] = None
        self._foreign_max = 0
        self._optimizer = None
        self.bf16_pull = self.dist and self.arena.shadow is not None and self.on_gpu
        if self.bf16_pull:
            self._build_small_table()

    # ------------------------------------------------------------------ layout
    def owned_ranges(self) -> Optional[List[Tuple[int, int]]]:
        if not self.dist:
            return None
        return [self.arena.shard_range(b, self.rank, self.world) for b in range(len(self.arena.buckets))]

    @staticmethod
    def consumed_in_fp32(p) -> bool:
        """True for a parameter some kernel reads from the fp32 MASTER: every tensor that has no bf16 shadow pair
        (``_ddl_shadow_t``: conv filters and Linear / LSTM weights with both sides % 8 == 0 are consumed through it).
        That is all 1-D tensors (BN affine, biases), the 3-channel stem filter, the Conv1d front of the LSTM workload,
        Linear weights with a ragged side.  They must be identical on every replica, so they travel exactly."""
        return getattr(p, "_ddl_shadow_t", None) is None

    def _build_small_table(self) -> None:
        """Index tables of the tensors consumed in fp32: positions in the arena, a 0/1 mask of the positions this rank
        owns (~63k elements for ResNet-50), and the table of the foreign ranges (everything other ranks own)."""
        a = self.arena
        idx, own = [], []
        ranges = self.owned_ranges() or []
        for p, s in zip(a.params, a.segments):
            if not self.consumed_in_fp32(p):
                continue
            lo, hi = ranges[s.bucket]
            r = torch.arange(s.offset, s.offset + s.numel, dtype=torch.int64)
            idx.append(r)
            own.append((r >= lo) & (r < hi))
        if idx:
            self._small_idx = torch.cat(idx).to(a.device)
            self._small_own = torch.cat(own).to(a.device)
            self._small_buf = torch.empty(self._small_idx.numel(), dtype=torch.float32, device=a.device)
        foreign = []
        for b, (lo, hi) in enumerate(ranges):
            bk = a.buckets[b]
            foreign += [r for r in ((bk.lo, lo), (hi, bk.hi)) if r[1] > r[0]]
        self._foreign = torch.tensor(foreign, dtype=torch.int64).reshape(-1, 2).to(a.device)
        self._foreign_max = max((hi - lo for lo, hi in foreign), default=0)

    # ------------------------------------------------------------------ push
    def _launch_collective(self, b: int, g: torch.Tensor) -> None:
        inv = 1.0 / self.world
        src = g if self.grad_dtype is None else self._pack(b, g)
        lo, hi = self.arena.shard_range(b, self.rank, self.world)
        dst = self.arena.grad[lo:hi]
        off = lo - self.arena.buckets[b].lo
        scale = False
        if self.ncomm is not None:
            part = dst if src is g else self._comm_buf(("shard", b), hi - lo)
            h = self.ncomm.reduce_scatter(part, src, average=True)
        elif self.backend == "nccl":
            part = dst if src is g else self._comm_buf(("shard", b), hi - lo)
            h = td.reduce_scatter_tensor(part, src, op=td.ReduceOp.AVG, group=self.pg, async_op=True)
        else:   # gloo has no reduce-scatter: same semantics via all-reduce, only the owned shard is used
            h = self._all_reduce_sum_async(src, ("bucket", b))
            part, scale = src[off:off + (hi - lo)], True

        def finish(dst=dst, part=part, scale=scale):
            if part.data_ptr() != dst.data_ptr() or part.dtype != dst.dtype:
                self._unpack(dst, part)
            if scale:
                dst.mul_(inv)

        self._handles.append((b, h, finish))

    # ------------------------------------------------------------------ pull
    def _all_gather(self, full: torch.Tensor, own: torch.Tensor):
        """``full`` = world x ``own`` in rank order (``own`` may alias its slot of ``full``).  Returns a waitable."""
        if self.ncomm is not None and own.dtype in (torch.float32, torch.bfloat16):
            return self.ncomm.all_gather(full, own)
        if self.backend == "nccl":
            return td.all_gather_into_tensor(full, own, group=self.pg, async_op=True)
        n = own.numel()
        if self.staged and own.is_cuda:
            host_full = torch.empty(full.numel(), dtype=full.dtype)
            views = [host_full[r * n:(r + 1) * n] for r in range(self.world)]
            td.all_gather(views, own.detach().cpu(), group=self.pg)
            full.copy_(host_full)
            return None
        views = [full[r * n:(r + 1) * n] for r in range(self.world)]
        return td.all_gather(views, own.clone(), group=self.pg, async_op=True)

    def gather_parameters(self) -> None:
        """Pull: every rank gets the updated parameters of all shards (see the module docstring for what travels)."""
        if not self.dist:
            return
        a = self.arena
        handles = []
        for b in range(len(a.buckets)):
            bk = a.buckets[b]
            lo, hi = a.shard_range(b, self.rank, self.world)
            if self.bf16_pull:
                handles.append(self._all_gather(a.shadow[bk.lo:bk.hi], a.shadow[lo:hi]))
            else:
                handles.append(self._all_gather(a.master[bk.lo:bk.hi], a.master[lo:hi]))
        for h in handles:
            if h is not None:
                h.wait()
        if self.bf16_pull:
            self._refresh_foreign_masters()
            a.refresh_transposes()
            a.mark_shadow_current()
        else:
            a.sync_shadow()

    def _refresh_foreign_masters(self) -> None:
        """fp32 masters of the ranges other ranks own: 1-D tensors exactly (one small fp32 all-reduce of the owned
        values, zero elsewhere), everything else from the gathered bf16 shadow."""
        a = self.arena
        from ..ops import kernels as K
        small = None
        if self._small_idx is not None:                       # before the cast below overwrites nothing we own: owned
            small = K.pack_owned(a.master, self._small_idx, self._small_own, self._small_buf)     # values, 0 elsewhere
            h = self.ncomm.all_reduce(small, average=False) if self.ncomm is not None else \
                self._all_reduce_sum_async(small, ("small",))
        if self._foreign_max > 0:                             # one launch for every foreign range, beside the all-reduce
            K.cast_ranges(a.shadow, a.master, self._foreign, self._foreign_max)
        if small is not None:
            if h is not None:
                h.wait()
            K.scatter_values(a.master, self._small_idx, small)

    def gather_master(self) -> None:
        """Exact fp32 masters of every shard on every rank (checkpoint time)."""
        if not self.dist:
            return
        a = self.arena
        handles = []
        for b in range(len(a.buckets)):
            bk = a.buckets[b]
            lo, hi = a.shard_range(b, self.rank, self.world)
            handles.append(self._all_gather(a.master[bk.lo:bk.hi], a.master[lo:hi]))
        for h in handles:
            if h is not None:
                h.wait()

    def gather_flat(self, compact: torch.Tensor) -> torch.Tensor:
        """A per-shard ("compact", owned ranges back to back in bucket order) fp32 buffer of every rank -> the full
        arena-layout buffer.  Collective; used for optimizer state when a checkpoint is written."""
        a = self.arena
        full = torch.zeros(a.total, dtype=compact.dtype, device=compact.device)
        if not self.dist:
            full.copy_(compact)
            return full
        off = 0
        handles = []
        for b in range(len(a.buckets)):
            bk = a.buckets[b]
            n = (bk.hi - bk.lo) // self.world
            handles.append(self._all_gather(full[bk.lo:bk.hi], compact[off:off + n]))
            off += n
        for h in handles:
            if h is not None:
                h.wait()
        return full

    def prepare_checkpoint(self) -> None:
        if self.dist and self.bf16_pull:
            self.gather_master()

    def _checksum_source(self) -> torch.Tensor:
        # foreign fp32 masters are bf16-rounded copies under the bf16 pull: compare what the replicas compute with
        return self.arena.shadow if self.bf16_pull else self.arena.master

    # ------------------------------------------------------------------ optimizer wiring
    def attach(self, optimizer) -> None:
        """Shard the optimizer: state only for the owned ranges, LARS norms summed over ranks, state gathered for
        checkpoints through this reducer."""
        self._optimizer = optimizer
        if not self.dist:
            return
        if hasattr(optimizer, "shard_state"):
            optimizer.shard_state(self.owned_ranges(), gather=self.gather_flat)
        if hasattr(optimizer, "norm_reduce"):
            if self.ncomm is not None:
                optimizer.norm_reduce = lambda norms: self.ncomm.all_reduce(norms, average=False).wait()
            elif self.staged:
                def reduce(norms):
                    host = norms.cpu()
                    td.all_reduce(host, op=td.ReduceOp.SUM, group=self.pg)
                    norms.copy_(host)
                optimizer.norm_reduce = reduce
            else:
                optimizer.norm_reduce = lambda norms: td.all_reduce(norms, op=td.ReduceOp.SUM, group=self.pg)

    def step(self, optimizer) -> None:
        """push (finish the reduce-scatters) -> sharded update -> pull."""
        self.finish_grad_sync()
        if self.on_synced is not None:
            self.on_synced()
        optimizer.step(ranges=self.owned_ranges())
        if self.dist:
            with device_of(self.arena.device):
                self.arena.zero_grad()      # non-owned gradient ranges were not cleared by the sharded step
                self.gather_parameters()

    def state_bytes_per_rank(self) -> Dict[str, int]:
        """Bytes of optimizer state this rank holds, per state tensor (= total / world once attached)."""
        opt = self._optimizer
        return {} if opt is None else {k: v.numel() * v.element_size() for k, v in opt.flat_state.items()}


def wrap(module, sync: str = "allreduce", **kw) -> DistributedDataParallel:
    """``sync`` = ``"allreduce"`` | ``"ps"`` (the ``--sync`` flag)."""
    kw.pop("mode", None)
    if sync == "ps":
        return ShardedDataParallel(module, **kw)
    return DistributedDataParallel(module, mode="allreduce", **kw)
