"""Batch loader: sampler -> index batches -> one gather per batch (device-resident data) or one async H2D per
batch from pinned memory, prefetched on a side HIP stream.  Replaces the reference's per-sample
``__getitem__`` + ``.to(device)`` + default collate path (``/root/reference/src/pytorch/CNN/main.py:177-179``)."""
from __future__ import annotations

from typing import Iterator, Optional, Tuple

import torch

from .sampler import ShardedSampler


class BatchLoader:
    def __init__(self, dataset, sampler: ShardedSampler, batch_size: int, device: Optional[torch.device] = None,
                 drop_last: bool = False, prefetch: bool = True):
        self.dataset, self.sampler, self.batch_size = dataset, sampler, int(batch_size)
        self.device = device
        self.drop_last = drop_last
        self.prefetch = prefetch and device is not None and device.type == "cuda"
        self._stream = torch.cuda.Stream(device) if self.prefetch else None

    def __len__(self) -> int:
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def set_epoch(self, epoch: int) -> None:
        self.sampler.set_epoch(epoch)

    def _fetch(self, idx: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if hasattr(self.dataset, "batch"):
            x, y = self.dataset.batch(idx)
        else:
            items = [self.dataset[int(i)] for i in idx]
            x = torch.stack([a for a, _ in items])
            y = torch.stack([torch.as_tensor(b) for _, b in items])
        if self.device is not None and x.device != self.device:
            if self.device.type == "cuda":
                x, y = x.pin_memory(), y.pin_memory()
            x = x.to(self.device, non_blocking=True)
            y = y.to(self.device, non_blocking=True)
        return x, y

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        order = torch.tensor(list(iter(self.sampler)), dtype=torch.int64)
        batches = [order[i:i + self.batch_size] for i in range(0, len(order), self.batch_size)]
        if self.drop_last and batches and len(batches[-1]) < self.batch_size:
            batches.pop()
        if not self.prefetch:
            for b in batches:
                yield self._fetch(b)
            return
        nxt = None
        cur_stream = torch.cuda.current_stream(self.device)
        for i in range(len(batches) + 1):
            cur = nxt
            if i < len(batches):
                with torch.cuda.stream(self._stream):
                    nxt = self._fetch(batches[i])
                    ev = torch.cuda.Event()
                    ev.record(self._stream)
                    nxt = (*nxt, ev)
            else:
                nxt = None
            if cur is not None:
                x, y, ev = cur
                cur_stream.wait_event(ev)
                x.record_stream(cur_stream)
                y.record_stream(cur_stream)
                yield x, y
