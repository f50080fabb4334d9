"""Metrics and log lines.

The four quoted log lines of the reference are kept verbatim (``/root/reference/src/pytorch/CNN/main.py:80,96,111,127``)
— epoch duration is still "ends at" minus "begins at" — with two fixes: the loss is a true per-sample mean (the
reference divides a sum of batch means by the sample count) and metrics can be all-reduced over ranks.
Accumulation stays on the device: no ``.item()`` per step (``main.py:92``); one host read per phase.
"""
from __future__ import annotations

import json
import os
import time
from datetime import datetime, timezone
from typing import Optional

import torch
import torch.distributed as td


def now_ts() -> float:
    """UNIX timestamp in UTC (the reference forces ``TZ=UTC``, ``main.py:23``)."""
    return datetime.now(timezone.utc).timestamp()


def fmt_train_begin(epoch: int, ts: float) -> str:
    return '"train epoch %d begins at %f"' % (epoch, ts)


def fmt_train_end(epoch: int, ts: float, acc: float, loss: float) -> str:
    return '"train epoch %d ends at %f with accuracy %0.03f and loss %0.09f"' % (epoch, ts, acc, loss)


def fmt_val_end(epoch: int, ts: float, acc: float, loss: float) -> str:
    return '"validation epoch %d ends at %f with accuracy %0.03f and loss %0.09f"' % (epoch, ts, acc, loss)


def fmt_test_end(ts: float, acc: float, loss: float) -> str:
    return '"test ends at %f with accuracy %0.03f and loss %0.09f"' % (ts, acc, loss)


class PhaseMeter:
    """Device-side accumulators for one phase (train / validation / test)."""

    def __init__(self, device: torch.device):
        self.acc = torch.zeros(3, dtype=torch.float64, device=device)   # loss*n, correct, n
        self.steps = 0
        self.t0 = time.perf_counter()

    def update(self, loss: torch.Tensor, correct: Optional[torch.Tensor], n: int) -> None:
        self.acc[0] += loss.detach().to(self.acc.dtype) * n
        if correct is not None:
            self.acc[1] += correct.detach().to(self.acc.dtype)
        self.acc[2] += n
        self.steps += 1

    def result(self, reduce: bool = False):
        """-> (accuracy %, mean loss, samples, seconds).  One host sync."""
        a = self.acc.clone()
        if reduce and td.is_available() and td.is_initialized() and td.get_world_size() > 1:
            from .dist import all_reduce_any
            all_reduce_any(a)                  # three doubles through the control plane (CPU-staged beside native RCCL)
        loss_n, correct, n = (float(v) for v in a.tolist())
        dt = time.perf_counter() - self.t0
        n = max(n, 1.0)
        return 100.0 * correct / n, loss_n / n, int(n), dt


class JsonlLogger:
    def __init__(self, path: Optional[str], enabled: bool = True):
        self.path = path if enabled else None
        if self.path:
            os.makedirs(os.path.dirname(os.path.abspath(self.path)), exist_ok=True)

    def log(self, **rec) -> None:
        if self.path:
            rec.setdefault("ts", now_ts())
            with open(self.path, "a") as f:
                f.write(json.dumps(rec) + "\n")


class StepTimer:
    """Step-time statistics from HIP events (GPU) or the host clock (CPU); percentiles without per-step syncs."""

    def __init__(self, device: torch.device, keep: int = 4096):
        self.gpu = device.type == "cuda"
        self.events, self.host, self.keep = [], [], keep
        self._last = None

    def tick(self) -> None:
        if self.gpu:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            if self._last is not None and len(self.events) < self.keep:
                self.events.append((self._last, ev))
            self._last = ev
        else:
            t = time.perf_counter()
            if self._last is not None and len(self.host) < self.keep:
                self.host.append((t - self._last) * 1e3)
            self._last = t

    def summary(self):
        if self.gpu:
            if not self.events:
                return {}
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in self.events)
        else:
            ms = sorted(self.host)
        if not ms:
            return {}
        pick = lambda q: ms[min(len(ms) - 1, int(q * len(ms)))]  # noqa: E731
        return {"steps": len(ms), "mean_ms": sum(ms) / len(ms), "p50_ms": pick(0.5), "p90_ms": pick(0.9),
                "p99_ms": pick(0.99), "max_ms": ms[-1]}
