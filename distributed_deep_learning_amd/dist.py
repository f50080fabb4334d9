"""Process-group bootstrap and rank discovery: one process per GPU.

Two planes.  The CONTROL plane is a ``torch.distributed`` process group: rendezvous, the 128-byte RCCL id, barriers,
metric and checksum reductions of a few bytes.  The DATA plane moves gradients and parameters: on GPUs the native RCCL
communicator (``parallel/native_comm.py``, one per device that holds parameters), on CPUs gloo.  With the native data
plane the control plane is GLOO even on GPUs: no second RCCL communicator inside c10d (double buffers, double init
time, one more tenant of the four hardware queues), and no process group bound to ONE device - which is what makes
data-parallel x model-parallel work, where a rank drives several GPUs and every stage has its own communicator
(``/root/reference/src/pytorch/CNN/main.py:134-154``).  ``backend="nccl"`` keeps everything inside ``torch.distributed``
(``--comm torch``).

Rank discovery honours, in order: explicit arguments, torchrun (``RANK/WORLD_SIZE/LOCAL_RANK``), Open MPI
(``OMPI_COMM_WORLD_*`` — what the reference reads at ``/root/reference/src/pytorch/CNN/main.py:62-67``), SLURM.
Unlike the reference nothing is hard-coded: no ``MASTER_ADDR='rtx2080-1.mit'`` / ``NCCL_SOCKET_IFNAME='enp3s0'``
(``main.py:24-27,192-193``); rendezvous comes from the launcher's environment.
"""
from __future__ import annotations

import datetime
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as td


@dataclass
class RankInfo:
    rank: int = 0
    world: int = 1
    local_rank: int = 0
    local_world: int = 1
    launcher: str = "none"     # none | torchrun | mpi | slurm | spawn

    @property
    def distributed(self) -> bool:
        return self.world > 1


def discover(env=None) -> RankInfo:
    e = os.environ if env is None else env
    if "RANK" in e and "WORLD_SIZE" in e:
        return RankInfo(int(e["RANK"]), int(e["WORLD_SIZE"]), int(e.get("LOCAL_RANK", 0)),
                        int(e.get("LOCAL_WORLD_SIZE", e["WORLD_SIZE"])), "torchrun")
    if "OMPI_COMM_WORLD_RANK" in e:
        return RankInfo(int(e["OMPI_COMM_WORLD_RANK"]), int(e.get("OMPI_COMM_WORLD_SIZE", 1)),
                        int(e.get("OMPI_COMM_WORLD_LOCAL_RANK", 0)), int(e.get("OMPI_COMM_WORLD_LOCAL_SIZE", 1)), "mpi")
    if "SLURM_PROCID" in e and "SLURM_NTASKS" in e:
        return RankInfo(int(e["SLURM_PROCID"]), int(e["SLURM_NTASKS"]), int(e.get("SLURM_LOCALID", 0)),
                        int(e.get("SLURM_NTASKS_PER_NODE", "1").split("(")[0]), "slurm")
    return RankInfo()


def backend_for(device: torch.device, comm: str = "native") -> str:
    """Control-plane backend: gloo on CPUs and beside the native RCCL data plane, nccl (= RCCL inside c10d) for
    ``comm="torch"`` on GPUs."""
    return "nccl" if (device.type == "cuda" and comm == "torch") else "gloo"


def init(info: Optional[RankInfo] = None, device: Optional[torch.device] = None, timeout_s: float = 600.0,
         master_addr: Optional[str] = None, master_port: Optional[int] = None, force: bool = False,
         comm: str = "native", bind_device: bool = True) -> RankInfo:
    """Initialise the default process group (no-op for world == 1).  Makes ``device`` the current device first so every
    RCCL communicator opens on the right GPU (the reference leaves every non-``data`` rank on ``cuda:0``).
    ``comm``: data plane on GPUs (see the module doc).  ``bind_device=False``: an nccl group that is not tied to one
    device (a rank that drives several GPUs: model / pipeline stages)."""
    info = info or discover()
    if device is not None and device.type == "cuda":
        torch.cuda.set_device(device)
    if (info.world <= 1 and not force) or td.is_initialized():     # force: a 1-rank group (diagnostics / tests)
        return info
    if master_addr is not None:
        os.environ["MASTER_ADDR"] = master_addr
    if master_port is not None:
        os.environ["MASTER_PORT"] = str(master_port)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = backend_for(device or torch.device("cpu"), comm)
    if backend == "gloo" and os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost", "::1") and os.path.isdir("/sys/class/net/lo"):
        # single-node rendezvous on the loopback address: gloo otherwise picks its interface by resolving the hostname,
        # which containers often cannot do
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    kwargs = {}
    if backend == "nccl" and device is not None and bind_device:
        kwargs["device_id"] = device
    td.init_process_group(backend=backend, rank=info.rank, world_size=info.world,
                          timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
    return info


def is_dist() -> bool:
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def rank() -> int:
    return td.get_rank() if is_dist() else 0


def world() -> int:
    return td.get_world_size() if is_dist() else 1


def barrier() -> None:
    if is_dist():
        td.barrier()


def shutdown() -> None:
    if td.is_available() and td.is_initialized():
        try:
            td.barrier()
        finally:
            td.destroy_process_group()


def control_device(group=None, fallback: Optional[torch.device] = None) -> torch.device:
    """Where a tensor must live to go through the control plane: CPU for gloo; for nccl the device the group is bound
    to, else ``fallback`` (an unbound nccl group takes tensors of any device)."""
    if not (td.is_available() and td.is_initialized()) or td.get_backend(group) != "nccl":
        return torch.device("cpu")
    pg = group if group is not None else td.distributed_c10d._get_default_group()   # noqa: SLF001
    bound = getattr(pg, "bound_device_id", None)
    if bound is not None:
        return torch.device(bound)
    return fallback if fallback is not None and fallback.type == "cuda" else torch.device("cuda", torch.cuda.current_device())


def all_reduce_any(t: torch.Tensor, op=None, group=None) -> torch.Tensor:
    """Small all-reduce of a tensor on ANY device through the control plane (metrics, flags, checksums): staged
    through the control plane's device when the tensor lives elsewhere.  In place; returns ``t``."""
    if not (td.is_available() and td.is_initialized()) or td.get_world_size(group) <= 1:
        return t
    op = td.ReduceOp.SUM if op is None else op
    dev = control_device(group, t.device)
    if dev == t.device:
        td.all_reduce(t, op=op, group=group)
    else:
        h = t.detach().to(dev)
        td.all_reduce(h, op=op, group=group)
        t.copy_(h)
    return t


def broadcast_flat(t: torch.Tensor, src: int = 0) -> None:
    if is_dist():
        dev = control_device(None, t.device)
        if dev == t.device:
            td.broadcast(t, src)
        else:
            h = t.detach().to(dev)
            td.broadcast(h, src)
            t.copy_(h)


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    return all_reduce_any(t)
