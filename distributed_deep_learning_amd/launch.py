"""Launcher: one process per rank.

* launched by an external launcher (torchrun / mpirun / srun): ranks are discovered from the environment
  (:func:`..dist.discover`) and this process is one rank — the reference's MPI path
  (``/root/reference/src/pytorch/CNN/main.py:62-67,190-196``);
* ``-m data -r N`` without a launcher: spawn N local ranks (``main.py:197-203``) with a private rendezvous on
  127.0.0.1 (free port picked per job; nothing hard-coded at import time like ``main.py:22-27``);
* otherwise: run in-process.

A failing rank raises in the parent (``spawn(join=True)``), which tears the job down; the process-group timeout
turns a hung collective into an error instead of a deadlock.  Recovery story: restart with ``--resume``.

    python -m distributed_deep_learning_amd --model cnn -m data -r 2 -e 2
"""
from __future__ import annotations

import os
import socket
import sys
from typing import List, Optional

import torch
import torch.multiprocessing as mp

from . import dist as ddist
from .config import Config, parse_args


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn_entry(local_rank: int, world: int, cfg: Config, port: int, result_path: Optional[str]):
    os.environ["MASTER_ADDR"] = cfg.master_addr
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(local_rank)
    os.environ["LOCAL_RANK"] = str(local_rank)
    os.environ["WORLD_SIZE"] = str(world)
    os.environ["LOCAL_WORLD_SIZE"] = str(world)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if cfg.device == "cpu":
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // world))
    from .trainer import run
    info = ddist.RankInfo(local_rank, world, local_rank, world, "spawn")
    cfg.master_port = port
    hist = run(cfg, info)
    if result_path and local_rank == 0:
        import json
        with open(result_path, "w") as f:
            json.dump(hist, f)
    ddist.shutdown()


def launch(cfg: Config, result_path: Optional[str] = None):
    info = ddist.discover()
    if info.world > 1:                                  # external launcher: this process is one rank
        from .trainer import run
        hist = run(cfg, info)
        ddist.shutdown()
        return hist
    if cfg.mode == "data" and cfg.run > 1:              # single node, spawn N ranks
        if cfg.device == "gpu" and torch.cuda.device_count() < cfg.run:
            raise RuntimeError(f"-r {cfg.run} needs {cfg.run} GPUs, found {torch.cuda.device_count()}")
        port = free_port()
        mp.spawn(_spawn_entry, args=(cfg.run, cfg, port, result_path), nprocs=cfg.run, join=True)
        return None
    from .trainer import run
    return run(cfg, info)


def main(argv: Optional[List[str]] = None) -> int:
    os.environ.setdefault("TZ", "UTC")
    launch(parse_args(argv))
    return 0


if __name__ == "__main__":
    sys.exit(main())
