"""Spawn helper for multi-process gloo tests (world_size > 1 on CPU, rendezvous on 127.0.0.1)."""
import os
import socket
import traceback

import torch
import torch.distributed as td
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _plain(o):
    """Tensors -> numpy so results are pickled by value (no shared-memory handles tied to the child's life)."""
    if isinstance(o, torch.Tensor):
        return ("__tensor__", o.detach().cpu().numpy())
    if isinstance(o, dict):
        return {k: _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_plain(v) for v in o)
    return o


def _unplain(o):
    if isinstance(o, tuple) and len(o) == 2 and isinstance(o[0], str) and o[0] == "__tensor__":
        return torch.from_numpy(o[1].copy())
    if isinstance(o, dict):
        return {k: _unplain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_unplain(v) for v in o)
    return o


def _entry(rank, world, port, fn, args, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank))
        torch.set_num_threads(2)
        td.init_process_group("gloo", rank=rank, world_size=world)
        out = fn(rank, world, *args)
        q.put((rank, "ok", _plain(out)))
        td.barrier()
        td.destroy_process_group()
    except Exception:  # noqa: BLE001
        q.put((rank, "err", traceback.format_exc()))
        raise


def run_ranks(fn, world=2, args=()):
    """Run ``fn(rank, world, *args)`` on ``world`` gloo ranks; returns the list of per-rank results."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_entry, args=(r, world, port, fn, args, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, status, out = q.get(timeout=240)
        if status != "ok":
            for p in procs:
                p.terminate()
            raise AssertionError(f"rank {rank} failed:\n{out}")
        results[rank] = _unplain(out)
    for p in procs:
        p.join(60)
    return [results[r] for r in range(world)]
