"""Process-wide state shared by the optimizers, the layer weight caches and the kernel wrappers.

The fused optimizers update master weights through raw device pointers, which torch's per-tensor version
counters cannot see; they call :func:`bump` after every step so cached bf16 filter repacks are refreshed.

:func:`set_deterministic` selects bit-reproducible weight gradients (``csrc/conv_wgrad.hip``: split-M partial tiles in
slabs + an ordered reduce instead of float atomics).  Also on with ``DDL_DETERMINISTIC=1`` or
``torch.use_deterministic_algorithms(True)``; ``Config.deterministic`` / ``--deterministic`` set it for a training run.
"""
import os

import torch

_weights_epoch = 0
_deterministic = os.environ.get("DDL_DETERMINISTIC", "0") == "1"


def set_deterministic(flag: bool) -> None:
    global _deterministic
    _deterministic = bool(flag)


def deterministic() -> bool:
    return _deterministic or torch.are_deterministic_algorithms_enabled()


def bump() -> int:
    global _weights_epoch
    _weights_epoch += 1
    return _weights_epoch


def weights_epoch() -> int:
    return _weights_epoch
