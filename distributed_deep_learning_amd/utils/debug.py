"""Debug-mode checks: the reference has none, and its spawn-mode data parallelism silently
trains UNSYNCHRONISED replicas (``/root/reference/src/pytorch/CNN/main.py:133-142``: ``sync`` stays a no-op).

* :func:`replica_divergence` — compares parameter checksums across ranks (what would have caught that bug);
* :func:`check_finite` — one fused reduction over the flat arenas instead of a per-tensor Python loop;
* :class:`GradLedger` — bookkeeping assertions for the gradient-ready notifications that drive the bucketed
  reducer: every segment fires exactly once per backward pass, and a bucket is never reduced before all of its
  segments fired (the ordering contract between compute streams and the communication stream).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch


def replica_divergence(ddp) -> float:
    """max over ranks of |checksum_rank - checksum_rank0| (0.0 = bit-identical replicas)."""
    cs = ddp.param_checksum()
    return float((cs - cs[0:1]).abs().max())


def check_finite(arena, what: str = "grad") -> bool:
    """True when every element of the arena's gradient (or ``"master"`` weights) is finite."""
    t = arena.grad if what == "grad" else arena.master
    return bool(torch.isfinite(t).all())


class GradLedger:
    """Attach with ``ledger = GradLedger(ddp)``; raises ``AssertionError`` at the first contract violation.

    ``snapshot=True`` adds a data-level check for single-rank runs (identity collectives): every bucket is cloned
    on the stream and at the point its collective is enqueued, and once the step's collectives have finished
    (``ddp.step`` -> ``on_synced``) the bucket must still hold exactly those values — a gradient kernel that landed
    after its bucket was reduced shows up as a difference.  This is what proves the readiness signals of the kernels
    that write straight into the arena (they bypass autograd's accumulation) on a one-GPU box."""

    def __init__(self, ddp, snapshot: bool = False):
        self.ddp = ddp
        self.signals: Dict[int, int] = {}            # every readiness signal, duplicates included
        self.fired: Dict[int, int] = {}              # accepted (first) signals: segment -> arrival index
        self.launched_at: Dict[int, int] = {}
        self.violations: List[str] = []
        self.snapshot = snapshot and ddp.world == 1 and ddp.grad_dtype is None
        self.snaps: Dict[int, torch.Tensor] = {}
        self.verified = 0
        self._orig_ready = ddp._segment_ready        # noqa: SLF001
        self._orig_launch = ddp._launch              # noqa: SLF001
        self._orig_collective = ddp._launch_collective   # noqa: SLF001
        self._orig_arm = ddp._arm                    # noqa: SLF001
        ddp._arm = self._arm                         # noqa: SLF001
        ddp._launch = self._launch                   # noqa: SLF001
        ddp._segment_ready = self._ready             # noqa: SLF001
        if self.snapshot:
            ddp._launch_collective = self._collective    # noqa: SLF001
            ddp.on_synced = self.verify_complete

    def _arm(self) -> None:
        self.signals.clear()
        self.fired.clear()
        self.launched_at.clear()
        self.snaps.clear()
        self._orig_arm()

    def _ready(self, seg: int) -> None:
        d = self.ddp
        if d._armed and d.require_sync:              # noqa: SLF001
            self.signals[seg] = self.signals.get(seg, 0) + 1
            if self.signals[seg] > 2:                # a native notify plus autograd's hook is the most one pass sends
                self._fail(f"segment {seg} signalled ready {self.signals[seg]} times in one backward pass")
            if not d._seen[seg]:                     # noqa: SLF001
                self.fired[seg] = len(self.fired)
        self._orig_ready(seg)

    def _launch(self, b: int) -> None:
        d = self.ddp
        pending = d._pending[b]                      # noqa: SLF001
        if pending > 0:
            self._fail(f"bucket {b} reduced with {pending} gradient(s) still outstanding")
        if pending < 0:
            self._fail(f"bucket {b} over-counted: {-pending} more readiness signals than segments")
        self.launched_at[b] = len(self.launched_at)
        self._orig_launch(b)

    def _collective(self, b: int, g: torch.Tensor) -> None:
        self.snaps[b] = g.clone()                    # same stream, same point in the queue as the collective
        self._orig_collective(b, g)

    def verify_complete(self) -> None:
        """Single-rank snapshot check (see the class docstring); a no-op unless ``snapshot`` is active."""
        if not self.snapshot:
            return
        for b, snap in self.snaps.items():
            now = self.ddp.arena.bucket_grad(b)
            if not torch.equal(now, snap):
                bad = int((now != snap).sum())
                self._fail(f"bucket {b}: {bad} gradient element(s) changed after its collective was enqueued")
            self.verified += 1

    def _fail(self, msg: str) -> None:
        self.violations.append(msg)
        raise AssertionError("gradient ledger: " + msg)

    def launch_order(self) -> List[int]:
        """Buckets in the order their collectives were enqueued (expected: last bucket first)."""
        return [b for b, _ in sorted(self.launched_at.items(), key=lambda kv: kv[1])]
