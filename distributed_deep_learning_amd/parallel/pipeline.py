"""Micro-batch pipeline parallelism on top of :class:`StagedModel`.

The batch is split into micro-batches of ``micro_batch`` samples (the reference's ``-p`` flag is the micro-batch
SIZE, ``/root/reference/src/pytorch/CNN/model.py:212``) and pushed through a GPipe-style fill / steady / drain
schedule: at clock tick ``t`` stage ``s`` works on micro-batch ``t - s``.  One host thread issues the work; since
HIP kernel launches and peer copies are asynchronous, stage ``s`` on GPU ``s`` overlaps stage ``s-1`` on GPU
``s-1``.  ``forward`` alone (evaluation, or the reference's own training loop) leaves backward to plain autograd over
the recorded graph of ALL micro-batches, as in the reference: every micro-batch's activations stay alive until the end.

``train_step`` is the 1F1B-style training schedule the reference does not have: a micro-batch's loss and backward pass
are issued in the clock tick in which it leaves the last stage, while the following micro-batches are still moving
forward through the earlier stages - at most ``stages`` micro-batches hold activations at any time instead of all of
them, and backward kernels of micro-batch ``c`` overlap forward kernels of ``c+1 ...`` on the other devices.  Gradients
accumulate in the stage arenas; data-parallel reducers communicate only behind the LAST micro-batch (``no_sync``
around the others).  ``schedule_1f1b`` lists the issue order and is unit-tested with ``schedule``.

The reference's schedule duplicates chunk 0 and drops chunk ``N-S`` whenever there are more chunks than stages
(``model.py:231-253``, SURVEY A4 — this includes its default settings); ``schedule()`` below is the corrected
order and is unit-tested for every (stages, chunks) pair.
"""
from __future__ import annotations

import contextlib
from typing import List, Sequence, Tuple

import torch

from ..ops._lib import device_of
from .mp import StagedModel


def schedule(n_chunks: int, n_stages: int) -> List[List[Tuple[int, int]]]:
    """Clock ticks of the forward pipeline: ``ticks[t]`` = [(stage, chunk), ...], later stages first, so a stage's
    input (produced in the previous tick) is consumed before the earlier stage overwrites its slot."""
    if n_chunks < 1 or n_stages < 1:
        raise ValueError("need at least one chunk and one stage")
    ticks = []
    for t in range(n_chunks + n_stages - 1):
        ticks.append([(s, t - s) for s in range(n_stages - 1, -1, -1) if 0 <= t - s < n_chunks])
    return ticks


def schedule_1f1b(n_chunks: int, n_stages: int) -> List[Tuple[str, int, int]]:
    """Issue order of one training step: ``("F", stage, chunk)`` and ``("B", -1, chunk)`` (the whole backward pass of a
    chunk, last stage first).  A chunk's ``B`` directly follows the tick in which its last ``F`` ran."""
    ops: List[Tuple[str, int, int]] = []
    for tick in schedule(n_chunks, n_stages):
        for s, c in tick:
            ops.append(("F", s, c))
        for s, c in tick:
            if s == n_stages - 1:
                ops.append(("B", -1, c))
    return ops


class PipelinedModel(StagedModel):
    def __init__(self, model, devices: Sequence, micro_batch: int, **kw):
        super().__init__(model, devices, **kw)
        if micro_batch is None or micro_batch < 1:
            raise ValueError("micro_batch (the pipeline size) must be >= 1")
        self.micro_batch = int(micro_batch)

    def forward(self, x):
        chunks = list(x.split(self.micro_batch, dim=0))
        n, s_count = len(chunks), len(self.stages)
        slot = [None] * s_count          # slot[s]: output of stage s waiting for stage s+1
        outputs = [None] * n
        for tick in schedule(n, s_count):
            for s, c in tick:
                inp = chunks[c] if s == 0 else slot[s - 1]
                out = self.run_stage(s, inp)
                if s == s_count - 1:
                    outputs[c] = out
                else:
                    slot[s] = out
        return torch.cat(outputs, dim=0)

    def train_step(self, x, y, criterion, reducers: Sequence = ()):
        """Forward + loss + backward of one batch on the 1F1B-style schedule (module docstring).  ``criterion`` returns
        the MEAN loss of what it is given; the step's loss is the sample-weighted mean over the micro-batches, so the
        accumulated gradients equal those of one whole-batch backward (up to BatchNorm, which sees micro-batches in
        either schedule).  Returns ``(loss, correct)`` - detached device tensors (``correct`` is ``None`` when the
        criterion does not count hits)."""
        chunks = list(x.split(self.micro_batch, dim=0))
        targets = list(y.split(self.micro_batch, dim=0))
        n, s_count, total = len(chunks), len(self.stages), x.shape[0]
        slot = [None] * s_count
        live = [r for r in reducers if r is not None]
        loss_sum, correct_sum = None, None
        self.max_live_chunks = 0                  # diagnostics: chunks holding activations at once
        in_flight = 0
        for tick in schedule(n, s_count):
            finished = None
            for s, c in tick:
                if s == 0:
                    in_flight += 1
                    self.max_live_chunks = max(self.max_live_chunks, in_flight)
                out = self.run_stage(s, chunks[c] if s == 0 else slot[s - 1])
                if s == s_count - 1:
                    finished = (c, out)
                else:
                    slot[s] = out
            if finished is None:
                continue
            c, out = finished
            with device_of(out.device):
                t = targets[c].to(out.device, non_blocking=True)
                loss = criterion(out, t) * (chunks[c].shape[0] / total)
                hits = getattr(criterion, "last_correct", None)
                with contextlib.ExitStack() as stack:
                    if c != n - 1:                # reducers talk behind the last micro-batch only
                        for r in live:
                            stack.enter_context(r.no_sync())
                    loss.backward()
                loss_sum = loss.detach() if loss_sum is None else loss_sum + loss.detach()
                if hits is not None:
                    correct_sum = hits if correct_sum is None else correct_sum + hits
            in_flight -= 1
            del out, loss
        return loss_sum, correct_sum
