"""LR schedules.  ``StepLR`` mirrors the reference (``StepLR(step_size=7, gamma=0.1)``,
``/root/reference/src/pytorch/CNN/main.py:161``); ``WarmupPolyLR`` / ``WarmupCosineLR`` are the usual large-batch
LARS schedules (north-star config 5).  All are plain ``torch.optim.lr_scheduler.LambdaLR`` factories, so they
(de)serialise with ``state_dict`` like any torch scheduler."""
from __future__ import annotations

import math

from torch.optim.lr_scheduler import LambdaLR, StepLR  # noqa: F401  (re-exported)


def WarmupPolyLR(optimizer, warmup_steps: int, total_steps: int, power: float = 2.0, end_factor: float = 0.0):
    def f(step: int) -> float:
        if warmup_steps > 0 and step < warmup_steps:
            return (step + 1) / warmup_steps
        t = min(max(step - warmup_steps, 0), max(total_steps - warmup_steps, 1))
        frac = 1.0 - t / max(total_steps - warmup_steps, 1)
        return end_factor + (1.0 - end_factor) * frac ** power
    return LambdaLR(optimizer, f)


def WarmupCosineLR(optimizer, warmup_steps: int, total_steps: int, end_factor: float = 0.0):
    def f(step: int) -> float:
        if warmup_steps > 0 and step < warmup_steps:
            return (step + 1) / warmup_steps
        t = min(max(step - warmup_steps, 0), max(total_steps - warmup_steps, 1)) / max(total_steps - warmup_steps, 1)
        return end_factor + (1.0 - end_factor) * 0.5 * (1.0 + math.cos(math.pi * t))
    return LambdaLR(optimizer, f)
