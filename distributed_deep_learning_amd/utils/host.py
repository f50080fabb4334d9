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


def gpu_count_no_hip() -> int:
    """Number of GPUs this process would see, WITHOUT initialising the HIP runtime: a launcher that is about to start
    one child per GPU must not create a HIP context itself (``torch.cuda.device_count()`` falls back to
    ``hipGetDeviceCount`` on ROCm builds without amdsmi).  Source: the KFD topology in sysfs (a GPU node has
    ``simd_count > 0``), narrowed by ``HIP_VISIBLE_DEVICES`` / ``ROCR_VISIBLE_DEVICES`` / ``CUDA_VISIBLE_DEVICES``.
    Returns -1 when sysfs has no KFD topology (no driver: the caller may fall back to asking a throw-away child)."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = sorted(os.listdir(root), key=lambda v: int(v) if v.isdigit() else 1 << 30)
    except OSError:
        return -1
    gpus = 0
    for n in nodes:
        try:
            with open(os.path.join(root, n, "properties")) as f:
                props = dict(line.split(None, 1) for line in f if " " in line)
        except OSError:
            continue
        try:
            if int(props.get("simd_count", "0").strip()) > 0:
                gpus += 1
        except ValueError:
            continue
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            gpus = min(gpus, len(ids))
    return gpus


def gpu_count_via_child() -> int:
    """Ask a throw-away child process for ``torch.cuda.device_count()`` (the parent never touches HIP)."""
    import subprocess
    import sys
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                           text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1])
    except Exception:       # noqa: BLE001
        return 0


def visible_gpus() -> int:
    n = gpu_count_no_hip()
    return n if n >= 0 else gpu_count_via_child()
