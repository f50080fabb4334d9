"""Synthetic datasets with the shapes of the reference's three workloads and of the north-star configs.

There is no network and none of the reference's data (``/data/PCB_DATASET``, ``/data/PredictiveMaintenance``,
``/data/MQTT``) exists here, so every dataset is generated: fixed-seed random inputs whose labels are a
deterministic function of the input (a fixed random linear "teacher"), so models can actually learn and tests can
assert that the loss goes down.  Data is held as whole tensors (optionally device-resident): batches are slices,
not per-sample ``__getitem__`` + ``.to(device)`` calls like ``/root/reference/src/pytorch/CNN/dataset.py:107``.
"""
from __future__ import annotations

from typing import Tuple

import torch


class TensorDataset(torch.utils.data.Dataset):
    """Map-style dataset over pre-built tensors with a fast ``batch(indices)`` path."""

    def __init__(self, x: torch.Tensor, y: torch.Tensor):
        if len(x) != len(y):
            raise ValueError("x / y length mismatch")
        self.x, self.y = x, y

    def __len__(self) -> int:
        return len(self.x)

    def __getitem__(self, i):
        return self.x[i], self.y[i]

    def batch(self, idx: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        idx = idx.to(self.x.device)
        return self.x.index_select(0, idx), self.y.index_select(0, idx)

    def to(self, device) -> "TensorDataset":
        self.x, self.y = self.x.to(device), self.y.to(device)
        return self


def _teacher_labels(x: torch.Tensor, classes: int, g: torch.Generator) -> torch.Tensor:
    flat = x.reshape(len(x), -1).float()
    w = torch.randn(flat.shape[1], classes, generator=g) / flat.shape[1] ** 0.5
    return (flat @ w).argmax(1)


def image_dataset(n: int, channels: int, height: int, width: int, classes: int, seed: int = 0,
                  one_hot: bool = False) -> TensorDataset:
    """Images ~ N(0,1); ImageNet / CIFAR / MNIST / PCB-crop shapes (``ImageShapeDataset`` of SURVEY C15)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, channels, height, width, generator=g)
    y = _teacher_labels(x, classes, g)
    if one_hot:
        y = torch.nn.functional.one_hot(y, classes).float()
    return TensorDataset(x, y)


def mnist_like(n=2048, seed=0):
    return image_dataset(n, 1, 28, 28, 10, seed)


def cifar_like(n=2048, seed=0):
    return image_dataset(n, 3, 32, 32, 10, seed)


def imagenet_like(n=256, seed=0, classes=1000):
    return image_dataset(n, 3, 224, 224, classes, seed)


def pcb_like(n=5906, seed=0, one_hot=True):
    """64x64 RGB crops, 6 defect classes, float one-hot targets: the reference CNN dataset's shape
    (``/root/reference/src/pytorch/CNN/dataset.py:100-108``; 5,906 samples after its 2x augmentation)."""
    return image_dataset(n, 3, 64, 64, 6, seed, one_hot=one_hot)


def tabular_dataset(n=8192, features=48, classes=5, seed=0, one_hot=True) -> TensorDataset:
    """MQTT-intrusion shape: 48 features -> 5 classes (``/root/reference/src/pytorch/MLP/dataset.py:33-37``)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, features, generator=g)
    y = _teacher_labels(x, classes, g)
    if one_hot:
        y = torch.nn.functional.one_hot(y, classes).float()
    return TensorDataset(x, y)


class WindowDataset(torch.utils.data.Dataset):
    """Predictive-maintenance shape: ``machines`` series of ``rows`` rows; a sample is a window of ``history``
    consecutive rows of one machine (never spanning machines) -> (history x features, targets of the window's
    FIRST row), as in ``/root/reference/src/pytorch/LSTM/dataset.py:36-45``."""

    def __init__(self, machines=8, rows=256, features=32, targets=5, history=10, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.machines, self.rows, self.history = machines, rows, history
        self.data = torch.randn(machines, rows, features, generator=g)
        w = torch.randn(features, targets, generator=g) / features ** 0.5
        self.targets = torch.tanh(self.data @ w)
        self.per_machine = rows - (history - 1)

    def __len__(self) -> int:
        return self.machines * self.per_machine

    def idx2pos(self, idx: int) -> Tuple[int, int]:
        m = idx // self.per_machine
        return m, idx - m * self.per_machine          # window = rows [start, start+history)

    def __getitem__(self, idx):
        m, s = self.idx2pos(int(idx))
        return self.data[m, s:s + self.history], self.targets[m, s]

    def batch(self, idx: torch.Tensor):
        idx = idx.to(self.data.device)
        m = torch.div(idx, self.per_machine, rounding_mode="floor")
        s = idx - m * self.per_machine
        ar = torch.arange(self.history, device=idx.device)
        rows = s[:, None] + ar[None, :]
        return self.data[m[:, None], rows], self.targets[m, s]

    def to(self, device):
        self.data, self.targets = self.data.to(device), self.targets.to(device)
        return self
