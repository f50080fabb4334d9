"""Native RCCL communicator (``csrc/comm.cpp``): collectives issued straight from this library on a HIP stream the
communicator owns, next to — not through — ``torch.distributed``.

Why: one ctypes call per collective instead of a c10d work object, a communication stream whose ordering against
the compute streams is explicit (events), and calls that are plain stream operations (capturable in a hipGraph).
``torch.distributed`` stays the bootstrap (the control plane of ``dist.py``): the 128-byte RCCL unique id travels
through the default process group - on its CONTROL device, so a gloo group never sees a GPU tensor and a device-bound nccl
group never sees a foreign device - or through a ``TCPStore``.  RCCL itself is the copy torch ships (``torch/lib/librccl.so``), resolved with
``dlopen`` so the process never maps two RCCLs.

Default transport of the GPU reducer (``DistributedDataParallel(comm="native")`` / ``--comm native``); the reducer
compares one all-reduce with ``torch.distributed``'s at start-up and every rank falls back together if they differ.
The stream has normal priority on purpose: a high-priority stream that waits on events of normal-priority streams is
the pattern behind the GPU-side deadlocks recorded in ``docs/STATUS.md``.

Tuning for the xGMI mesh (7 point-to-point links per GPU) is RCCL's environment, passed through untouched:
``NCCL_MIN_NCHANNELS`` / ``NCCL_MAX_NCHANNELS`` (channels, i.e. links driven at once), ``NCCL_ALGO`` (``Ring`` is
per-link bound: 2(N-1)/N x S / 153 GB/s; ``Tree`` / direct algorithms use more links for small N),
``NCCL_PROTO``, ``NCCL_BUFFSIZE``; the bucket size is ``--bucket-mb``.

Reference behaviour replaced: blocking per-parameter ``all_reduce`` through c10d
(``/root/reference/src/pytorch/CNN/main.py:134-140``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from ..ops import _lib
from ..ops._lib import check, device_of

_DT = {torch.float32: 0, torch.bfloat16: 1}


def _rccl_path() -> Optional[bytes]:
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p.encode() if os.path.exists(p) else None


class _Handle:
    """Completion of one collective: ``wait()`` orders the CURRENT stream behind it (no host blocking)."""

    __slots__ = ("event", "device")

    def __init__(self, event: torch.cuda.Event, device=None):
        self.event = event
        self.device = device

    def wait(self) -> None:
        torch.cuda.current_stream(self.device).wait_event(self.event)


class NativeCommunicator:
    _store_seq = 0

    def __init__(self, rank: int, world: int, device: torch.device, group=None, store=None):
        """Collective constructor (every rank calls it).  The unique id is broadcast through ``group`` (a
        ``torch.distributed`` group, default: the default group when initialised) or exchanged through ``store``
        (anything with ``set`` / ``get``, e.g. ``torch.distributed.TCPStore``); a single rank needs neither."""
        self.lib = _lib.require()
        check(self.lib.ddl_comm_load(_rccl_path()), "ddl_comm_load (RCCL not found)")
        self.rank, self.world, self.device = rank, world, device
        self.profile = None
        self._t0 = self._tag = None
        self.proxy_workgroups = 0              # one-rank diagnostics: footprint of a ring kernel behind every all-reduce
        with device_of(device):             # ncclCommInitRank binds the CALLING THREAD's current device; restored after
            self._init(rank, world, device, group, store)

    def _init(self, rank, world, device, group, store) -> None:
        ident = (C.c_char * 128)()
        if rank == 0:
            check(self.lib.ddl_comm_unique_id(C.addressof(ident)), "ncclGetUniqueId")
        if world > 1:
            import torch.distributed as td
            if store is not None:
                # every rank opens its communicators in the same order: the sequence number keys the id of THIS one
                key = f"ddl_rccl_id/{NativeCommunicator._store_seq}"
                NativeCommunicator._store_seq += 1
                if rank == 0:
                    store.set(key, bytes(ident))
                else:
                    C.memmove(ident, store.get(key), 128)
            elif td.is_available() and td.is_initialized():
                from ..dist import control_device
                t = torch.frombuffer(bytearray(bytes(ident)), dtype=torch.uint8).clone()
                dev_t = t.to(control_device(group, device))
                td.broadcast(dev_t, 0, group=group)
                C.memmove(ident, bytes(dev_t.cpu().numpy().tobytes()), 128)
            else:
                raise RuntimeError("NativeCommunicator: world > 1 needs a process group or a store for the unique id")
        comm = C.c_void_p()
        check(self.lib.ddl_comm_init(C.addressof(ident), rank, world, C.byref(comm)), "ncclCommInitRank")
        self._comm = comm
        # a stream measured to overlap BOTH the compute stream and the wgrad side stream: HIP multiplexes streams onto 4
        # hardware queues round-robin, and a collective that lands on the compute queue would not overlap backward
        from ..ops import streams
        beside = [streams.side_stream(device)] if streams.enabled() else []
        self.stream = streams.pick_stream(device, beside=beside)

    @property
    def version(self) -> int:
        return self.lib.ddl_comm_version()

    def ranks(self) -> int:
        """Number of ranks the RCCL communicator itself reports (``ncclCommCount``)."""
        return int(self.lib.ddl_comm_count(self._comm))

    def user_rank(self) -> int:
        return int(self.lib.ddl_comm_user_rank(self._comm))

    # ------------------------------------------------------------------ plumbing
    def _enter(self, tag=None) -> int:
        """Order the comm stream behind everything enqueued on the current stream; returns its raw handle."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        if self.profile is not None:
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record(self.stream)
            self._tag = tag
        return self.stream.cuda_stream

    def _done(self) -> _Handle:
        if self.profile is not None:                          # diagnostics: time every collective on the comm stream
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(self.stream)
            self.profile.append((self._t0, ev, self._tag))
            return _Handle(ev, self.device)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return _Handle(ev, self.device)

    def start_profile(self) -> None:
        """From now on every collective is bracketed by timing events on the communicator's stream."""
        self.profile = []

    def stop_profile(self):
        """-> [(tag, milliseconds)] of the collectives since :meth:`start_profile` (host-blocking)."""
        rec, self.profile = self.profile or [], None
        out = []
        for t0, t1, tag in rec:
            t1.synchronize()
            out.append((tag, t0.elapsed_time(t1)))
        return out

    @staticmethod
    def _dt(t: torch.Tensor) -> int:
        if t.dtype not in _DT or not t.is_cuda or not t.is_contiguous():
            raise TypeError("native collectives take contiguous fp32 / bf16 GPU tensors")
        return _DT[t.dtype]

    # ------------------------------------------------------------------ collectives (asynchronous)
    def all_reduce(self, t: torch.Tensor, average: bool = True, tag=None) -> _Handle:
        with device_of(self.device):
            st = self._enter(tag if tag is not None else ("all_reduce", t.numel() * t.element_size()))
            check(self.lib.ddl_comm_all_reduce(self._comm, t.data_ptr(), t.data_ptr(), t.numel(), self._dt(t),
                                               int(average), st), "ncclAllReduce")
            if self.proxy_workgroups > 0 and self.world == 1:
                check(self.lib.ddl_stream_proxy(t.data_ptr(), t.numel() * t.element_size(), int(self.proxy_workgroups), 2,
                                                st), "ddl_stream_proxy")
            return self._done()

    def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor, average: bool = True, tag=None) -> _Handle:
        if inp.numel() != out.numel() * self.world or inp.dtype != out.dtype:
            raise ValueError("reduce_scatter: input must hold world x output elements of the same dtype")
        with device_of(self.device):
            st = self._enter(tag if tag is not None else ("reduce_scatter", inp.numel() * inp.element_size()))
            check(self.lib.ddl_comm_reduce_scatter(self._comm, inp.data_ptr(), out.data_ptr(), out.numel(),
                                                   self._dt(out), int(average), st), "ncclReduceScatter")
            return self._done()

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor) -> _Handle:
        if out.numel() != inp.numel() * self.world or inp.dtype != out.dtype:
            raise ValueError("all_gather: output must hold world x input elements of the same dtype")
        with device_of(self.device):
            st = self._enter(("all_gather", out.numel() * out.element_size()))
            check(self.lib.ddl_comm_all_gather(self._comm, inp.data_ptr(), out.data_ptr(), inp.numel(), self._dt(inp),
                                               st), "ncclAllGather")
            return self._done()

    def broadcast(self, t: torch.Tensor, root: int = 0) -> _Handle:
        with device_of(self.device):
            st = self._enter(("broadcast", t.numel() * t.element_size()))
            check(self.lib.ddl_comm_broadcast(self._comm, t.data_ptr(), t.numel(), self._dt(t), root, st), "ncclBroadcast")
            return self._done()

    def barrier(self) -> None:
        """All ranks have enqueued everything before this call AND it has finished on every GPU: a one-element
        all-reduce on the communicator's stream, then a host wait for it."""
        if getattr(self, "_bar", None) is None:
            self._bar = torch.zeros(8, dtype=torch.float32, device=self.device)
        self.all_reduce(self._bar, average=False)
        self.stream.synchronize()

    def abort(self) -> None:
        """Give up on outstanding collectives (``ncclCommAbort``): a rank leaving a hung job."""
        if getattr(self, "_comm", None):
            self.lib.ddl_comm_abort(self._comm)
            self._comm = None

    def close(self) -> None:
        if getattr(self, "_comm", None):
            self.stream.synchronize()
            self.lib.ddl_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 - interpreter shutdown
            pass
