"""Dataset splitting and rank sharding.

``split_indices``: seeded random permutation cut 70/10/20 (train/val/test) like the reference
(``/root/reference/src/pytorch/CNN/main.py:165-171``).

``ShardedSampler``: shards *the given subset's own indices* across ranks — the reference wraps a
``SubsetRandomSampler`` in a ``DistributedSampler`` (``main.py:173-175``), which silently ignores the subset and
makes train/val/test overlap.  Here every rank gets a disjoint, equally long slice of the
subset; ``set_epoch`` reshuffles identically on all ranks; the tail is padded (wrap-around) or dropped.
"""
from __future__ import annotations

import math
from typing import Iterator, List, Sequence, Tuple

import torch


def split_indices(n: int, fractions: Sequence[float] = (0.7, 0.1, 0.2), seed: int = 42) -> Tuple[torch.Tensor, ...]:
    if n < 0 or any(f < 0 for f in fractions) or sum(fractions) > 1.0 + 1e-9:
        raise ValueError("bad split fractions")
    g = torch.Generator().manual_seed(int(seed))
    perm = torch.randperm(n, generator=g)
    cuts, start = [], 0
    for i, f in enumerate(fractions):
        end = n if (i == len(fractions) - 1 and abs(sum(fractions) - 1.0) < 1e-9) else start + int(n * f)
        cuts.append(perm[start:end])
        start = end
    return tuple(cuts)


class ShardedSampler(torch.utils.data.Sampler):
    def __init__(self, indices, rank: int = 0, world: int = 1, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = False, pad: bool = True):
        """``pad=True`` (training): every rank gets ``ceil(n / world)`` indices, the tail wraps around, so all ranks run
        the same number of steps (their collectives must pair up).  ``pad=False`` (evaluation): no index is repeated —
        ranks get ``n // world`` or ``n // world + 1`` indices — so metrics summed over ranks count every sample of
        the split exactly once."""
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} outside world {world}")
        self.indices = torch.as_tensor(indices, dtype=torch.int64).clone()
        self.rank, self.world = rank, world
        self.shuffle, self.seed, self.drop_last, self.pad = shuffle, int(seed), drop_last, bool(pad)
        self.epoch = 0
        n = len(self.indices)
        if drop_last:
            self.num_samples = n // world
        elif self.pad:
            self.num_samples = math.ceil(n / world)
        else:
            self.num_samples = n // world + (1 if rank < n % world else 0)
        self.total = n if (not drop_last and not self.pad) else self.num_samples * world

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def _order(self) -> torch.Tensor:
        idx = self.indices
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = idx[torch.randperm(len(idx), generator=g)]
        if self.drop_last:
            return idx[: self.total]
        if self.total > len(idx) and len(idx) > 0:
            reps = math.ceil(self.total / len(idx))
            idx = idx.repeat(reps)[: self.total]
        return idx

    def __iter__(self) -> Iterator[int]:
        return iter(self._order()[self.rank:self.total:self.world].tolist())

    def __len__(self) -> int:
        return self.num_samples

    def state_dict(self):
        return {"epoch": self.epoch, "seed": self.seed}

    def load_state_dict(self, sd):
        self.epoch, self.seed = int(sd["epoch"]), int(sd["seed"])
