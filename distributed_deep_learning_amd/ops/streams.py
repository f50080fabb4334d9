"""Side HIP stream for the filter-gradient (wgrad) kernels.

A wgrad only feeds the optimizer / the gradient all-reduce: nothing else in the backward chain waits for it.  Run
on a second HIP stream it overlaps the memory-bound BN-backward passes and the next layer's dgrad on the main
stream (MFMA-bound work next to HBM-bound work on the same CUs) instead of sitting serially between them.

Ordering rules, all enforced here:

* fork: the side stream waits for the main stream before every wgrad (its operands were just produced there);
* lifetime: the operands are kept ALIVE by this module until the side-stream kernel that reads them has finished (an
  event per fork, polled at the next fork) or until the join — NOT ``Tensor.record_stream``: with record_stream the
  caching allocator defers every such block's reuse to its own event polling, the main stream runs ahead of it, and
  the pool answers with fresh ``hipMalloc`` calls in steady state (measured on ResNet-50, B=512: 200-320 device
  allocations inside 20 timed steps, 135 GB reserved for a ~45 GB working set, and in a quarter of all runs one step
  stalled for 1.2-2.0 s — `docs/STATUS.md`); that policy is gone, there is no switch that brings it back;
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
_held: dict[int, list] = {}          # device -> [(event or None, operands)] still (possibly) read on the side stream


def enabled() -> bool:
    return _ENABLED


def set_enabled(flag: bool) -> None:
    global _ENABLED
    join()
    _ENABLED = bool(flag)


def _overlaps(main: torch.cuda.Stream, cand: torch.cuda.Stream, us: int = 150) -> bool:
    """True when a kernel on ``cand`` runs CONCURRENTLY with one on ``main``: two single-wave kernels that each stay
    resident for ``us`` microseconds finish in ~1x that time on different hardware queues and in ~2x on a shared one."""
    from . import _lib
    lib = _lib.require()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cand.wait_stream(main)
    a.record(main)
    lib.ddl_spin(us, main.cuda_stream)
    lib.ddl_spin(us, cand.cuda_stream)
    main.wait_stream(cand)
    b.record(main)
    b.synchronize()
    return a.elapsed_time(b) * 1e3 < 1.6 * us


def _lib_error():
    from . import _lib
    return (_lib.HipKernelError, RuntimeError)


def _stream(index: int) -> torch.cuda.Stream:
    s = _side.get(index)
    if s is None:
        # HIP deals streams round-robin onto GPU_MAX_HW_QUEUES (4) hardware queues; once other libraries (RCCL, MIOpen)
        # have created theirs, a new stream can land on the MAIN stream's queue, where it overlaps nothing (seen in
        # rocprof: same queue id, kernel-time sum == busy time, -2 ms per step).  So the candidates are MEASURED and the
        # first one that really runs beside the current stream is kept.
        # Always NORMAL priority.  A high-priority stream gets a hardware queue of its own by construction, but a
        # fork/join between a high-priority and a normal-priority stream deadlocks on the GPU as soon as the process
        # holds more streams than hardware queues (bench/probes/prio_stream_probe.py: {normal, high} x {0, 8 extra
        # streams} - only high+8 hangs, within the first 500 iterations; MIOpen's RNN and gloo's GPU collectives put a
        # process into that regime, which is how round 1 met it).  There is no switch to turn that back on.
        main = torch.cuda.current_stream(index)
        cands = []
        for _ in range(5):
            c = torch.cuda.Stream(device=index, priority=0)
            cands.append(c)
            try:
                if torch.cuda.is_current_stream_capturing() or _overlaps(main, c):
                    break
            except _lib_error():                    # the calibration kernel could not be launched: keep the candidate
                break
        s = cands[-1]
        _side[index] = s
    return s


def pick_stream(device: torch.device, beside=(), tries: int = 8) -> torch.cuda.Stream:
    """A new normal-priority stream that was MEASURED to run concurrently with the current stream and with every stream
    in ``beside`` (the communicator's stream: a collective that shares a hardware queue with the compute stream or with
    the wgrad side stream serialises with it instead of overlapping backward).  Falls back to the last candidate when
    no candidate passes (or while a capture is in progress / the calibration kernel cannot be launched)."""
    main = torch.cuda.current_stream(device)
    cand = None
    for _ in range(max(1, tries)):
        cand = torch.cuda.Stream(device=device, priority=0)
        try:
            if torch.cuda.is_current_stream_capturing():
                break
            if _overlaps(main, cand) and all(_overlaps(o, cand) for o in beside):
                break
        except _lib_error():
            break
    return cand


def side_stream(device: torch.device) -> torch.cuda.Stream:
    """The wgrad side stream of ``device`` (created and calibrated on first use)."""
    with torch.cuda.device(device):
        return _stream(device.index if device.index is not None else torch.cuda.current_device())


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
    capturing = torch.cuda.is_current_stream_capturing()
    held = _held.setdefault(dev.index, [])
    if not capturing:
        while held and held[0][0] is not None and held[0][0].query():    # side-stream work finished: operands may go
            held.pop(0)
    _dirty.add(dev.index)
    if not _callback_queued:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)   # noqa: SLF001
            _callback_queued = True
        except RuntimeError:        # not inside a backward pass: the caller joins explicitly
            pass
    with torch.cuda.stream(side):
        yield
        ev = None
        if not capturing:
            ev = torch.cuda.Event()
            ev.record(side)
    held.append((ev, operands))


def join() -> None:
    """Make the current stream of every device with outstanding side-stream work wait for it."""
    global _callback_queued
    _callback_queued = False            # also recovers from a backward pass that died before its callback ran
    for index in list(_dirty):
        torch.cuda.current_stream(index).wait_stream(_side[index])
        # everything enqueued on the current stream from here on is ordered behind the side stream: the operands'
        # memory may be reused by it
        _held.pop(index, None)
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
