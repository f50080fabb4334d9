"""Device-agnostic data-parallel workers: the SAME checks run on gloo / CPU (everywhere) and on RCCL with one GPU per
rank (tests/test_multi_gpu.py).  Every worker builds its tensors on ``_dist_util.device()``; a world of 1 forces the
(identity) collectives so the transport is still exercised."""
import os

import torch
import torch.nn as tnn

from _dist_util import device


def mlp(seed):
    torch.manual_seed(seed)
    return tnn.Sequential(tnn.Linear(12, 40), tnn.Tanh(), tnn.Linear(40, 24), tnn.Tanh(), tnn.Linear(24, 3)).to(device())


def data(rank, n=8):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(n, 12, generator=g).to(device()), torch.randn(n, 3, generator=g).to(device())


def _ddp(m, world, **kw):
    from distributed_deep_learning_amd.parallel import DistributedDataParallel
    if device().type == "cuda":
        kw.setdefault("comm", os.environ.get("DDL_TEST_COMM", "native"))
    return DistributedDataParallel(m, force_collectives=world == 1, **kw)


def grad_mean_worker(rank, world, mode, bucket_mb, grad_dtype=None):
    """Synchronised gradients of one step (different initial weights per rank: the broadcast must fix that)."""
    from distributed_deep_learning_amd.parallel import ParamArena
    m = mlp(7 + rank)
    arena = ParamArena(m.named_parameters(), bucket_bytes=int(bucket_mb * 2**20), shard_multiple=world)
    ddp = _ddp(m, world, mode=mode, arena=arena, grad_dtype=grad_dtype)
    x, y = data(rank)
    ((ddp(x) - y) ** 2).mean().backward()
    ddp.finish_grad_sync()
    torch.cuda.synchronize() if device().type == "cuda" else None
    out = {"grad": ddp.arena.grad.clone(), "nb": len(ddp.arena.buckets), "ranges": ddp.owned_ranges(),
           "w": ddp.arena.master.clone(),
           "transport": "native" if ddp.ncomm is not None else ddp.backend,
           "comm_ranks": ddp.comm_ranks()}
    ddp.close()
    return out


def reference_mean_grad(world, bucket_mb):
    """Single process, CPU: mean over ranks of the per-rank gradients with rank 0's weights."""
    from distributed_deep_learning_amd.parallel import ParamArena
    torch.manual_seed(7)
    m = tnn.Sequential(tnn.Linear(12, 40), tnn.Tanh(), tnn.Linear(40, 24), tnn.Tanh(), tnn.Linear(24, 3))
    arena = ParamArena(m.named_parameters(), bucket_bytes=int(bucket_mb * 2**20), shard_multiple=world)
    grads = []
    for r in range(world):
        arena.zero_grad()
        g = torch.Generator().manual_seed(100 + r)
        x, y = torch.randn(8, 12, generator=g), torch.randn(8, 3, generator=g)
        ((m(x) - y) ** 2).mean().backward()
        grads.append(arena.grad.clone())
    return sum(grads) / world, arena.total


def train_worker(rank, world, mode, opt_kind, steps):
    from distributed_deep_learning_amd.optim import LARS, FusedSGD
    m = mlp(3)
    ddp = _ddp(m, world, bucket_mb=2e-3, mode=mode)
    if opt_kind == "lars":
        opt = LARS(ddp.arena, lr=0.5, momentum=0.9, weight_decay=1e-3, eta=0.02)
    else:
        opt = FusedSGD(ddp.arena, lr=0.05, momentum=0.9, weight_decay=1e-3)
    ddp.attach(opt)
    for s in range(steps):
        x, y = data(rank * 1000 + s)
        ((ddp(x) - y) ** 2).mean().backward()
        ddp.step(opt)
    cs = ddp.param_checksum()
    out = {"w": torch.cat([p.detach().flatten().float() for p in m.parameters()]), "checksum": cs}
    ddp.close()
    return out


def no_sync_worker(rank, world):
    m = mlp(9)
    ddp = _ddp(m, world)
    x, y = data(rank)
    with ddp.no_sync():
        ((ddp(x) - y) ** 2).mean().backward()
    local = ddp.arena.grad.clone()
    ((ddp(x) - y) ** 2).mean().backward()               # second micro-batch: accumulate, then sync
    ddp.finish_grad_sync()
    out = (local, ddp.arena.grad.clone())
    ddp.close()
    return out


def unused_param_worker(rank, world):
    torch.manual_seed(0)
    m = tnn.ModuleDict({"a": tnn.Linear(4, 4), "b": tnn.Linear(4, 4)}).to(device())
    ddp = _ddp(m, world, bucket_mb=1e-4)
    ddp._arm()                                           # noqa: SLF001
    m["a"](torch.ones(2, 4, device=device()) * (rank + 1)).sum().backward()      # "b" never gets a gradient
    ddp.finish_grad_sync()                                                       # must not deadlock
    out = ddp.arena.grad.clone()
    ddp.close()
    return out


def hybrid_worker(rank, world, mode, stages, batch, dev_kind):
    """DP x model / pipeline parallel through the Trainer: `stages` devices per rank (GPU: cuda:{rank*stages + i})."""
    from distributed_deep_learning_amd.config import Config
    from distributed_deep_learning_amd.trainer import Trainer
    cfg = Config(model="mlp", mode=mode, stages=stages, epochs=2, batch=batch, samples=320, optimizer="sgd", lr=0.05,
                 global_metrics=True, graph=False, pipeline=4, device=dev_kind)
    t = Trainer(cfg, printer=lambda *_: None).setup()
    hist = t.fit()
    sd = t.model.state_dict()
    out = {"w": torch.cat([v.detach().flatten().float().cpu() for _, v in sorted(sd.items())]),
           "loss": [h["train_loss"] for h in hist if "train_loss" in h],
           "n_reducers": sum(d is not None for d in t.ddps), "n_arenas": len(t.arenas),
           "devices": [str(d) for d in t.devices],
           "transports": ["native" if (d is not None and d.ncomm is not None) else "torch" for d in t.ddps]}
    for d in t.ddps:
        if d is not None:
            d.close()
    return out
