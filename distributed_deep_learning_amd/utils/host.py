"""Host-side runtime hygiene for launch-bound training loops.

A process that has imported torch and built a model tracks ~270k long-lived Python objects.  CPython's generational
collector re-scans ALL of them on every full (generation-2) collection — measured 67-115 ms on the MI355X host
(``bench/probes/gc_stall_probe.py``) — and a training step allocates enough container objects (autograd nodes, ctx
tuples) to trigger one every few dozen steps.  For a step that the host enqueues in 3 ms and the GPU runs in 4.4 ms
(ResNet-18/CIFAR, the reference's CNN) that pause starves the GPU: one step in ~50 took 85-115 ms, +40 % on the mean.

:func:`freeze_gc` collects once and moves every surviving object to the permanent generation (``gc.freeze``), so later
full collections only look at what the steps themselves allocate (< 1 ms).  Called by the Trainer after set-up and by
``bench.py`` after warm-up.  ``DDL_GC_FREEZE=0`` disables it.
"""
from __future__ import annotations

import gc
import os

_ENABLED = os.environ.get("DDL_GC_FREEZE", "1") != "0"


def freeze_gc() -> int:
    """Collect garbage, then freeze the survivors.  Returns the number of frozen objects (0 when disabled)."""
    if not _ENABLED:
        return 0
    gc.collect()
    gc.freeze()
    return gc.get_freeze_count()


def unfreeze_gc() -> None:
    """Undo :func:`freeze_gc` (tests; long-lived processes that rebuild their models)."""
    gc.unfreeze()
