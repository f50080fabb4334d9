"""Spawn helper for multi-process tests, rendezvous on 127.0.0.1.

``backend="gloo"``: CPU tensors, any world size (the tier that runs everywhere).
``backend="rccl"``: rank r owns ``cuda:r``; the process group is the framework's own bootstrap
(``distributed_deep_learning_amd.dist.init``: gloo control plane + native RCCL data plane, or an nccl group for
``comm="torch"``).  Tests of this tier carry ``@pytest.mark.gpu`` and ``skipif(visible_gpus() < world)``: they skip with
a reason on a 1-GPU box and run unchanged on an 8-GPU node.  Workers find their device through :func:`device`."""
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


def device() -> torch.device:
    """The device of the calling test rank (``cpu`` under gloo, ``cuda:LOCAL_RANK`` under rccl)."""
    return torch.device(os.environ.get("DDL_TEST_DEVICE", "cpu"))


def visible_gpus() -> int:
    """GPUs on this machine, counted without initialising HIP in the pytest process."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from distributed_deep_learning_amd.utils.host import visible_gpus as count
    return max(0, count())


def _entry(rank, world, port, fn, args, q, backend="gloo", comm="native"):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.set_num_threads(2)
        if backend == "rccl":
            from distributed_deep_learning_amd import dist as ddist
            dev = torch.device(f"cuda:{rank}")
            os.environ["DDL_TEST_DEVICE"] = str(dev)
            os.environ["DDL_TEST_COMM"] = comm
            ddist.init(ddist.RankInfo(rank, world, rank, world, "spawn"), dev, comm=comm, force=True, timeout_s=240.0)
        else:
            os.environ["DDL_TEST_DEVICE"] = "cpu"
            td.init_process_group("gloo", rank=rank, world_size=world)
        out = fn(rank, world, *args)
        q.put((rank, "ok", _plain(out)))
        td.barrier()
        td.destroy_process_group()
    except Exception:  # noqa: BLE001
        q.put((rank, "err", traceback.format_exc()))
        raise


def run_ranks(fn, world=2, args=(), backend="gloo", comm="native"):
    """Run ``fn(rank, world, *args)`` on ``world`` ranks (gloo / CPU or rccl / one GPU each); returns the per-rank results.
    The rendezvous port is found by binding port 0 and releasing it, so another process can take it before rank 0
    listens on it (seen once on the GPU box: EADDRINUSE): that launch is repeated on a fresh port, up to three times."""
    for attempt in range(3):
        try:
            return _run_ranks_once(fn, world, args, backend, comm)
        except AssertionError as e:
            if attempt == 2 or "EADDRINUSE" not in str(e):
                raise
    raise AssertionError("unreachable")


def _run_ranks_once(fn, world, args, backend, comm):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_entry, args=(r, world, port, fn, args, q, backend, comm)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, status, out = q.get(timeout=240)
        if status != "ok":
            for p in procs:
                p.terminate()
            for p in procs:
                p.join(30)
            raise AssertionError(f"rank {rank} failed:\n{out}")
        results[rank] = _unplain(out)
    for p in procs:
        p.join(60)
    return [results[r] for r in range(world)]
