"""Inter-layer model parallelism: partition a model's stages over devices and run them in sequence, moving the
activation between devices (a peer copy over xGMI on MI355X).

One general partitioner replaces the reference's three heuristics (``i//4`` in
``/root/reference/src/pytorch/CNN/model.py:196-201``, which breaks for >2 devices — SURVEY A12 — plus the MLP and
LSTM variants in ``MLP/model.py:62-76`` and ``LSTM/model.py:98-124``).  The reference tables are still available as
``reference_partition_mlp`` / ``reference_partition_lstm`` / ``reference_partition_cnn`` (compatibility mode,
golden values in SURVEY §5.7.3).

Any device list works — including repeated CPU devices, which is how the logic is tested without GPUs (the
reference's own trick, ``/root/reference/src/pytorch/LSTM/model.py:183``).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn as tnn

from ..ops._lib import device_of


# ----------------------------------------------------------------------------------------------- partitioning
def partition_layers(n_layers: int, n_stages: int, weights: Optional[Sequence[float]] = None) -> List[int]:
    """Contiguous partition of ``n_layers`` into ``n_stages`` groups minimising the heaviest group (exact DP).
    Returns ``stage_of_layer``.  Every stage gets at least one layer when ``n_layers >= n_stages``."""
    if n_layers < 1 or n_stages < 1:
        raise ValueError("need at least one layer and one stage")
    n_stages = min(n_stages, n_layers)
    w = [1.0] * n_layers if weights is None else [float(x) for x in weights]
    if len(w) != n_layers:
        raise ValueError("weights length mismatch")
    pre = [0.0]
    for x in w:
        pre.append(pre[-1] + x)
    INF = float("inf")
    # best[k][i]: minimal max-load splitting the first i layers into k stages
    best = [[INF] * (n_layers + 1) for _ in range(n_stages + 1)]
    cut = [[0] * (n_layers + 1) for _ in range(n_stages + 1)]
    best[0][0] = 0.0
    for k in range(1, n_stages + 1):
        for i in range(k, n_layers - (n_stages - k) + 1):
            for j in range(k - 1, i):
                cand = max(best[k - 1][j], pre[i] - pre[j])
                if cand < best[k][i] - 1e-12:
                    best[k][i], cut[k][i] = cand, j
    out = [0] * n_layers
    i = n_layers
    for k in range(n_stages, 0, -1):
        j = cut[k][i]
        for l in range(j, i):
            out[l] = k - 1
        i = j
    return out


def _expand(sizes: Sequence[int]) -> Dict[int, int]:
    out, l = {}, 0
    for s, n in enumerate(sizes):
        for _ in range(n):
            out[l] = s
            l += 1
    return out


def reference_partition_cnn(n_layers: int, n_devices: int) -> Dict[int, int]:
    """Compatibility table of the reference CNN: everything on device 0, or four logical layers per device."""
    return {i: (0 if n_devices == 1 else i // 4) for i in range(n_layers)}


def reference_partition_mlp(n_layers: int, n_devices: int) -> Dict[int, int]:
    """Compatibility table of the reference MLP (``layers = hidden + 2``): even split; one spare layer goes to the
    first device when more than one is spare, the remaining spares to the last devices."""
    base, spare = divmod(n_layers, n_devices)
    head = 1 if spare > 1 else 0
    tail = spare - head
    sizes = [base] * n_devices
    sizes[0] += head
    for d in range(n_devices - tail, n_devices):
        sizes[d] += 1
    return _expand(sizes)


def reference_partition_lstm(n_layers: int, n_devices: int) -> Dict[int, int]:
    """Compatibility table of the reference LSTM (``layers = hidden + 3``: conv, pool, LSTMs..., fc)."""
    if n_layers == n_devices:
        return {i: i for i in range(n_layers)}
    n_lstm = n_layers - 3
    base = n_lstm // n_devices
    spare = n_lstm - base * n_devices
    out = {0: 0}
    dev, room = (1, 1) if base < 1 else (0, base)
    for layer in range(2, n_lstm + 2):
        out[layer] = dev
        room -= 1
        if room < 1:
            dev += 1
            room = base + (1 if spare > 0 else 0)
            spare -= 1 if spare > 0 else 0
    out[n_layers - 1] = min(n_devices - 1, max(out.values()) + 1)
    out[1] = (out[2] - out[0]) // 2
    return dict(sorted(out.items()))


# ----------------------------------------------------------------------------------------------- staging
class FnStage(tnn.Module):
    """A stage defined by a callable plus the modules it owns (so ``.to(device)`` moves the right parameters)."""

    def __init__(self, fn: Callable, *owned: tnn.Module):
        super().__init__()
        self.fn = fn
        self.owned = tnn.ModuleList([m for m in owned if isinstance(m, tnn.Module)])

    def forward(self, x):
        return self.fn(x)


def stage_modules(model: tnn.Module) -> List[tnn.Module]:
    """Ordered, self-contained stage modules of a model: ``model.stage_modules()`` if defined, the children of an
    ``nn.Sequential``, else the model itself as a single stage."""
    if hasattr(model, "stage_modules"):
        return list(model.stage_modules())
    if isinstance(model, tnn.Sequential):
        return list(model)
    return [model]


def _weight(m: tnn.Module) -> float:
    return 1.0 + sum(p.numel() for p in m.parameters())


class StagedModel(tnn.Module):
    """``model`` split into ``len(devices)`` pipeline stages; forward = run stage i on ``devices[i]``."""

    def __init__(self, model, devices: Sequence, partition: Optional[Sequence[int]] = None,
                 weights: Optional[Sequence[float]] = None):
        super().__init__()
        if not isinstance(devices, (list, tuple)) or len(devices) < 1:
            raise TypeError("The devices argument must be a non-empty list of devices")
        self.devices = [torch.device(d) for d in devices]
        mods = stage_modules(model) if isinstance(model, tnn.Module) else list(model)
        if partition is None:
            partition = partition_layers(len(mods), len(self.devices), weights or [_weight(m) for m in mods])
        elif isinstance(partition, dict):
            partition = [partition[i] for i in range(len(mods))]
        if len(partition) != len(mods) or any(b < a for a, b in zip(partition, partition[1:])):
            raise ValueError("partition must be a non-decreasing stage index per layer")
        if max(partition) >= len(self.devices):
            raise ValueError(f"partition uses stage {max(partition)} but only {len(self.devices)} devices given")
        self.partition = list(partition)
        self.model = model if isinstance(model, tnn.Module) else None
        groups: List[List[tnn.Module]] = [[] for _ in self.devices]
        for m, s in zip(mods, partition):
            groups[s].append(m.to(self.devices[s]))
        self.stages = tnn.ModuleList(tnn.Sequential(*g) for g in groups)

    def run_stage(self, i: int, x):
        """Stage ``i`` on ``devices[i]``: the peer copy and every kernel of the stage are issued with that GPU as the
        current device, so the hand-written kernels (which launch on the current device's stream) run where their
        operands live, ordered with the ATen work and the copy on that device's stream."""
        dev = self.devices[i]
        with device_of(dev):
            if isinstance(x, torch.Tensor) and x.device != dev:
                x = x.to(dev, non_blocking=True)
            return self.stages[i](x)

    def forward(self, x):
        for i in range(len(self.stages)):
            x = self.run_stage(i, x)
        return x

    @property
    def in_device(self):
        return self.devices[0]

    @property
    def out_device(self):
        return self.devices[-1]

    # checkpoints keep the un-staged model's keys when one was given
    def state_dict(self, *args, **kwargs):
        if self.model is not None and not hasattr(self.model, "stage_modules") and isinstance(self.model, tnn.Sequential):
            return self.model.state_dict(*args, **kwargs)
        if self.model is not None:
            return self.model.state_dict(*args, **kwargs)
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, sd, *args, **kwargs):
        if self.model is not None:
            return self.model.load_state_dict(sd, *args, **kwargs)
        return super().load_state_dict(sd, *args, **kwargs)
