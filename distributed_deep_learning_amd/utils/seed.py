"""Deterministic seeding.  The reference seeds only torch (``/root/reference/src/pytorch/CNN/main.py:133``,
``torch.manual_seed(42)``) and relies on equal seeds for equal replicas; here equal replicas come from a rank-0
broadcast (``parallel/ddp.py``), and the seed additionally drives Python / NumPy and, optionally, a per-rank offset
for data augmentation streams."""
from __future__ import annotations

import os
import random

import numpy as np
import torch


def seed_everything(seed: int = 42, rank: int = 0, per_rank: bool = False) -> int:
    s = int(seed) + (int(rank) if per_rank else 0)
    random.seed(s)
    np.random.seed(s % (2 ** 32))
    torch.manual_seed(s)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(s)
    os.environ.setdefault("PYTHONHASHSEED", str(s))
    return s
