"""Side HIP stream for the filter-gradient (wgrad) kernels.

A wgrad only feeds the optimizer / the gradient all-reduce: nothing else in the backward chain waits for it.  Run
on a second HIP stream it overlaps the memory-bound BN-backward passes and the next layer's dgrad on the main
stream (MFMA-bound work next to HBM-bound work on the same CUs) instead of sitting serially between them.

Ordering rules, all enforced here:

* fork: the side stream waits for the main stream before every wgrad (its operands were just produced there);
* lifetime: operands are ``record_stream``-ed so the caching allocator does not hand their memory to a main-stream
  kernel while the wgrad still reads it;
* join: the first fork inside a backward pass queues an autograd-engine callback that makes the main stream wait
  for the side stream when the pass ends, so ``loss.backward()`` keeps its usual meaning (gradients are ordered
  before anything later on the current stream); :func:`join` does the same on demand (optimizer step);
* collectives: :func:`comm_context` lets the data-parallel reducer enqueue a bucket's all-reduce behind BOTH
  streams without stalling the main one.

``DDL_WGRAD_STREAM=0`` keeps everything on the current stream.
"""
from __future__ import annotations

import contextlib
import os

import torch

_ENABLED = os.environ.get("DDL_WGRAD_STREAM", "1") != "0"
_side: dict[int, torch.cuda.Stream] = {}
_dirty: set[int] = set()
_callback_queued = False


def enabled() -> bool:
    return _ENABLED


def set_enabled(flag: bool) -> None:
    global _ENABLED
    join()
    _ENABLED = bool(flag)


def _stream(index: int) -> torch.cuda.Stream:
    s = _side.get(index)
    if s is None:
        # A stream of a DIFFERENT priority class gets a hardware queue of its own.  Normal-priority streams are dealt
        # round-robin onto GPU_MAX_HW_QUEUES (4) queues, and once RCCL has created its streams the side stream can
        # land on the main stream's queue, where it overlaps nothing (measured: rocprof queue ids, kernel-time sum ==
        # busy time).  DDL_WGRAD_STREAM_PRIORITY: -1 high (default), 0 normal.
        prio = int(os.environ.get("DDL_WGRAD_STREAM_PRIORITY", "-1"))
        s = torch.cuda.Stream(device=index, priority=prio)
        _side[index] = s
    return s


def _end_of_backward() -> None:
    global _callback_queued
    _callback_queued = False
    join()


@contextlib.contextmanager
def fork(*operands: torch.Tensor):
    """``with fork(dy, x): launch_wgrad(...)`` — the launch lands on the side stream, ordered after everything the
    current stream has enqueued so far."""
    global _callback_queued
    dev = operands[0].device
    if not (_ENABLED and dev.type == "cuda"):
        yield
        return
    main = torch.cuda.current_stream(dev)
    side = _stream(dev.index)
    side.wait_stream(main)
    for t in operands:
        t.record_stream(side)
    _dirty.add(dev.index)
    if not _callback_queued:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)   # noqa: SLF001
            _callback_queued = True
        except RuntimeError:        # not inside a backward pass: the caller joins explicitly
            pass
    with torch.cuda.stream(side):
        yield


def join() -> None:
    """Make the current stream of every device with outstanding side-stream work wait for it."""
    for index in list(_dirty):
        torch.cuda.current_stream(index).wait_stream(_side[index])
    _dirty.clear()


@contextlib.contextmanager
def comm_context(t: torch.Tensor):
    """Enqueue a collective on ``t`` behind the main AND the side stream (the reducer's bucket launch)."""
    if not (_ENABLED and t.device.type == "cuda" and t.device.index in _dirty):
        yield
        return
    side = _side[t.device.index]
    side.wait_stream(torch.cuda.current_stream(t.device))
    with torch.cuda.stream(side):
        yield
