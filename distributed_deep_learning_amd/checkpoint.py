"""Checkpoint / resume (absent from the reference — SURVEY §5.4 — required by the north star).

The model part is a plain ``state_dict`` with stock PyTorch / torchvision key names, fp32, logical NCHW shapes:
a checkpoint written here loads into ``torchvision.models.resnet50()`` and vice versa (the channels-last / KRSC
physical layout and the bf16 shadow are internal).  Rank 0 writes atomically (tmp + rename); every rank loads,
then the DP wrapper re-broadcasts so replicas cannot diverge.  Optimizer state is stored flat with a name->range
layout table, so it can be restored under a different bucket plan or world size.  In the PS / ZeRO-1 mode the caller
runs ``reducer.prepare_checkpoint()`` (exact fp32 masters of every shard on every rank) and ``optimizer.state_dict()``
(a collective that all-gathers the momentum / Adam shards) on EVERY rank before rank 0 writes — see ``Trainer.save``.
"""
from __future__ import annotations

import os
import random
from typing import Any, Dict, Optional

import torch
import torch.distributed as td

FORMAT_VERSION = 1


def _rank() -> int:
    return td.get_rank() if td.is_available() and td.is_initialized() else 0


def _barrier() -> None:
    if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
        td.barrier()


def _dropout_state() -> Dict[str, Any]:
    """[seed, counter] of the fused-dropout generators (``ops.kernels.rng_state``); empty when no kernel drew a mask."""
    from .ops import kernels
    return kernels.dropout_state_dict()


def model_state(model) -> Dict[str, torch.Tensor]:
    return {k: v.detach().to("cpu").contiguous().clone() for k, v in model.state_dict().items()}


def save(path: str, model, optimizer=None, scheduler=None, epoch: int = 0, step: int = 0, sampler=None,
         extra: Optional[Dict[str, Any]] = None, all_ranks: bool = False) -> None:
    """Atomic save by rank 0 (``all_ranks=True`` writes ``path.rankN`` files, e.g. for per-rank RNG state)."""
    payload = {
        "format": FORMAT_VERSION,
        "model": model_state(model),
        "optimizer": optimizer.state_dict() if optimizer is not None else None,
        "scheduler": scheduler.state_dict() if scheduler is not None else None,
        "epoch": int(epoch),
        "step": int(step),
        "sampler": sampler.state_dict() if sampler is not None and hasattr(sampler, "state_dict") else None,
        "rng": {"torch": torch.get_rng_state(), "python": random.getstate(),
                "cuda": torch.cuda.get_rng_state_all() if torch.cuda.is_available() else None,
                "dropout": _dropout_state()},
        "extra": extra or {},
    }
    r = _rank()
    if r == 0 or all_ranks:
        target = path if r == 0 or not all_ranks else f"{path}.rank{r}"
        os.makedirs(os.path.dirname(os.path.abspath(target)), exist_ok=True)
        tmp = f"{target}.tmp.{os.getpid()}"
        torch.save(payload, tmp)
        os.replace(tmp, target)
    _barrier()


def load(path: str, model, optimizer=None, scheduler=None, sampler=None, strict: bool = True,
         restore_rng: bool = False) -> Dict[str, Any]:
    """Load on every rank.  Returns the metadata dict (``epoch``, ``step``, ``extra``)."""
    # weights_only: the payload is tensors, numbers, strings, lists, dicts and tuples only, so nothing from the file is
    # ever executed (a checkpoint can come from an untrusted place)
    payload = torch.load(path, map_location="cpu", weights_only=True)
    if payload.get("format") != FORMAT_VERSION:
        raise ValueError(f"unsupported checkpoint format {payload.get('format')}")
    model.load_state_dict(payload["model"], strict=strict)
    for p in model.parameters():
        arena = getattr(p, "_ddl_arena", None)
        if arena is not None:
            arena.sync_shadow()
            break
    if hasattr(model, "sync_parameters"):
        model.sync_parameters()
    if optimizer is not None and payload.get("optimizer") is not None:
        optimizer.load_state_dict(payload["optimizer"])
    if scheduler is not None and payload.get("scheduler") is not None:
        scheduler.load_state_dict(payload["scheduler"])
    if sampler is not None and payload.get("sampler") is not None and hasattr(sampler, "load_state_dict"):
        sampler.load_state_dict(payload["sampler"])
    if restore_rng and payload.get("rng"):
        torch.set_rng_state(payload["rng"]["torch"])
        random.setstate(payload["rng"]["python"])
        if payload["rng"].get("cuda") is not None and torch.cuda.is_available():
            torch.cuda.set_rng_state_all(payload["rng"]["cuda"])
    if (payload.get("rng") or {}).get("dropout"):      # always: the fused-dropout mask sequence continues where it stopped
        from .ops import kernels
        kernels.load_dropout_state_dict(payload["rng"]["dropout"])          # counters; every rank keeps its own seed
    _barrier()
    return {"epoch": payload["epoch"], "step": payload["step"], "extra": payload.get("extra", {})}


def load_model_state(path: str) -> Dict[str, torch.Tensor]:
    """Just the tensors, with a loader that executes nothing from the file."""
    payload = torch.load(path, map_location="cpu", weights_only=True)
    return payload["model"] if "model" in payload else payload
