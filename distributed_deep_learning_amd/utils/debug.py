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
    """Attach with ``ledger = GradLedger(ddp)``; raises ``AssertionError`` at the first contract violation."""

    def __init__(self, ddp):
        self.ddp = ddp
        self.fired: Dict[int, int] = {}
        self.launched_at: Dict[int, int] = {}
        self.violations: List[str] = []
        arena = ddp.arena
        self._seg_bucket = {i: s.bucket for i, s in enumerate(arena.segments)}
        self._orig_ready = ddp._param_ready          # noqa: SLF001
        self._orig_launch = ddp._launch              # noqa: SLF001
        self._orig_arm = ddp._arm                    # noqa: SLF001
        ddp._arm = self._arm                         # noqa: SLF001
        ddp._launch = self._launch                   # noqa: SLF001
        ddp._param_ready = self._ready               # noqa: SLF001
        self._rewire(arena)

    def _rewire(self, arena) -> None:
        seg_bucket = self._seg_bucket
        if arena.on_grad_ready is not None:
            arena.on_grad_ready = lambda seg: self._ready(seg_bucket[seg], seg)

    def _arm(self) -> None:
        self.fired.clear()
        self.launched_at.clear()
        self._orig_arm()

    def _ready(self, bucket: int, seg: Optional[int] = None) -> None:
        if seg is not None:
            self.fired[seg] = self.fired.get(seg, 0) + 1
            if self.fired[seg] > 1 and self.ddp.require_sync:
                self._fail(f"segment {seg} signalled ready {self.fired[seg]} times in one backward pass")
        self._orig_ready(bucket)

    def _launch(self, b: int) -> None:
        pending = self.ddp._pending[b]               # noqa: SLF001
        if pending > 0:
            self._fail(f"bucket {b} reduced with {pending} gradient(s) still outstanding")
        self.launched_at[b] = len(self.launched_at)
        self._orig_launch(b)

    def _fail(self, msg: str) -> None:
        self.violations.append(msg)
        raise AssertionError("gradient ledger: " + msg)

    def launch_order(self) -> List[int]:
        """Buckets in the order their collectives were enqueued (expected: last bucket first)."""
        return [b for b, _ in sorted(self.launched_at.items(), key=lambda kv: kv[1])]
