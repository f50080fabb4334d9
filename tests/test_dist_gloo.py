"""Multi-process data-parallel tests on CPU / gloo — the same reducer logic that runs on RCCL."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.nn as tnn

from _dist_util import run_ranks

pytestmark = pytest.mark.dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mlp(seed):
    torch.manual_seed(seed)
    return tnn.Sequential(tnn.Linear(12, 40), tnn.Tanh(), tnn.Linear(40, 24), tnn.Tanh(), tnn.Linear(24, 3))


def _data(rank, n=8):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(n, 12, generator=g), torch.randn(n, 3, generator=g)


# ---------------------------------------------------------------------------------------------
def _grad_mean_worker(rank, world, mode, bucket_mb):
    from distributed_deep_learning_amd.parallel import DistributedDataParallel, ParamArena
    m = _mlp(7 + rank)                                   # different init per rank: broadcast must fix it
    arena = ParamArena(m.named_parameters(), bucket_bytes=int(bucket_mb * 2**20), shard_multiple=world)
    ddp = DistributedDataParallel(m, mode=mode, arena=arena)
    x, y = _data(rank)
    ((ddp(x) - y) ** 2).mean().backward()
    ddp.finish_grad_sync()
    flat = ddp.arena.grad.clone()
    rng = ddp.owned_ranges()
    return {"grad": flat, "nb": len(ddp.arena.buckets), "ranges": rng,
            "w0": ddp.arena.master.clone() if rank == 0 else None, "w": ddp.arena.master.clone()}


@pytest.mark.parametrize("bucket_mb", [1e-3, 32.0])
def test_bucketed_allreduce_equals_mean_of_rank_grads(bucket_mb):
    out = run_ranks(_grad_mean_worker, 2, ("allreduce", bucket_mb))
    assert torch.equal(out[0]["w"], out[1]["w"])                 # rank-0 broadcast (reference defect A6)
    assert torch.equal(out[0]["grad"], out[1]["grad"])
    if bucket_mb < 1:
        assert out[0]["nb"] > 1
    # single-process reference: mean of the two per-rank gradients with rank 0's weights
    from distributed_deep_learning_amd.parallel import ParamArena
    m = _mlp(7)
    arena = ParamArena(m.named_parameters(), bucket_bytes=int(bucket_mb * 2**20), shard_multiple=2)
    grads = []
    for r in range(2):
        arena.zero_grad()
        x, y = _data(r)
        ((m(x) - y) ** 2).mean().backward()
        grads.append(arena.grad.clone())
    assert torch.allclose(out[0]["grad"], (grads[0] + grads[1]) / 2, atol=1e-7)


@pytest.mark.parametrize("world,mode", [(4, "ps"), (8, "allreduce"), (8, "ps")])
def test_reducer_at_the_benchmark_world_sizes(world, mode):
    """The same bucket / shard arithmetic at the world sizes of the scaling benchmark (4, 8 ranks): synchronised
    gradients equal the mean of the per-rank gradients, PS shards tile every bucket exactly."""
    out = run_ranks(_grad_mean_worker, world, (mode, 1e-3))
    from distributed_deep_learning_amd.parallel import ParamArena
    m = _mlp(7)
    arena = ParamArena(m.named_parameters(), bucket_bytes=int(1e-3 * 2**20), shard_multiple=world)
    grads = []
    for r in range(world):
        arena.zero_grad()
        x, y = _data(r)
        ((m(x) - y) ** 2).mean().backward()
        grads.append(arena.grad.clone())
    mean = sum(grads) / world
    assert out[0]["nb"] > 1
    for r in range(world):
        assert torch.equal(out[r]["w"], out[0]["w"])
        if mode == "allreduce":
            assert torch.allclose(out[r]["grad"], mean, atol=1e-6)
        else:
            for lo, hi in out[r]["ranges"]:
                assert torch.allclose(out[r]["grad"][lo:hi], mean[lo:hi], atol=1e-6)
    if mode == "ps":
        covered = sorted(x for r in range(world) for x in out[r]["ranges"])
        assert covered[0][0] == 0 and covered[-1][1] == arena.total
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))


def test_ps_mode_reduces_owned_shards():
    out = run_ranks(_grad_mean_worker, 2, ("ps", 1e-3))
    ref = run_ranks(_grad_mean_worker, 2, ("allreduce", 1e-3))
    for r in range(2):
        for lo, hi in out[r]["ranges"]:
            assert torch.allclose(out[r]["grad"][lo:hi], ref[r]["grad"][lo:hi], atol=1e-7)
    covered = sorted(x for r in range(2) for x in out[r]["ranges"])
    assert covered[0][0] == 0 and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))


# ---------------------------------------------------------------------------------------------
def _train_worker(rank, world, mode, opt_kind, steps):
    from distributed_deep_learning_amd.optim import LARS, FusedSGD
    from distributed_deep_learning_amd.parallel import DistributedDataParallel
    m = _mlp(3)
    ddp = DistributedDataParallel(m, bucket_mb=2e-3, mode=mode)
    if opt_kind == "lars":
        opt = LARS(ddp.arena, lr=0.5, momentum=0.9, weight_decay=1e-3, eta=0.02)
    else:
        opt = FusedSGD(ddp.arena, lr=0.05, momentum=0.9, weight_decay=1e-3)
    ddp.attach(opt)
    for s in range(steps):
        x, y = _data(rank * 1000 + s)
        ((ddp(x) - y) ** 2).mean().backward()
        ddp.step(opt)
    cs = ddp.param_checksum()
    return {"w": torch.cat([p.detach().flatten() for p in m.parameters()]), "checksum": cs, "names": ddp.arena.names}


@pytest.mark.parametrize("opt_kind", ["sgd", "lars"])
def test_replicas_identical_and_ps_equals_allreduce(opt_kind):
    ar = run_ranks(_train_worker, 2, ("allreduce", opt_kind, 4))
    ps = run_ranks(_train_worker, 2, ("ps", opt_kind, 4))
    assert torch.equal(ar[0]["w"], ar[1]["w"])                   # replicas bit-identical after k steps
    assert torch.equal(ps[0]["w"], ps[1]["w"])
    assert torch.equal(ar[0]["checksum"][0], ar[0]["checksum"][1])
    assert torch.allclose(ar[0]["w"], ps[0]["w"], atol=1e-6)     # PS mode == all-reduce mode
    assert ar[0]["w"].abs().sum() > 0


def _dp_vs_big_batch_worker(rank, world):
    from distributed_deep_learning_amd.optim import FusedSGD
    from distributed_deep_learning_amd.parallel import DistributedDataParallel
    m = _mlp(5)
    ddp = DistributedDataParallel(m)
    opt = FusedSGD(ddp.arena, lr=0.1, momentum=0.9)
    for s in range(3):
        x, y = _data(10 * s + rank)
        ((ddp(x) - y) ** 2).mean().backward()
        ddp.step(opt)
    return ddp.arena.master.clone()


def test_dp_world2_equals_single_process_double_batch():
    out = run_ranks(_dp_vs_big_batch_worker, 2)
    from distributed_deep_learning_amd.optim import FusedSGD
    m = _mlp(5)
    opt = FusedSGD(m.named_parameters(), lr=0.1, momentum=0.9)
    for s in range(3):
        xs, ys = zip(*[_data(10 * s + r) for r in range(2)])
        x, y = torch.cat(xs), torch.cat(ys)
        ((m(x) - y) ** 2).mean().backward()
        opt.step()
    assert torch.allclose(out[0], opt.arena.master, atol=1e-6)


def _no_sync_worker(rank, world):
    from distributed_deep_learning_amd.parallel import DistributedDataParallel
    m = _mlp(9)
    ddp = DistributedDataParallel(m)
    x, y = _data(rank)
    with ddp.no_sync():
        ((ddp(x) - y) ** 2).mean().backward()
    local = ddp.arena.grad.clone()
    ((ddp(x) - y) ** 2).mean().backward()               # second micro-batch: accumulate, then sync
    ddp.finish_grad_sync()
    return local, ddp.arena.grad.clone()


def test_no_sync_accumulates_then_reduces():
    out = run_ranks(_no_sync_worker, 2)
    assert not torch.equal(out[0][0], out[1][0])                 # unsynced local grads differ
    assert torch.equal(out[0][1], out[1][1])
    assert torch.allclose(out[0][1], out[0][0] + out[1][0], atol=1e-6)   # mean of (2*g0, 2*g1)


def _unused_param_worker(rank, world):
    from distributed_deep_learning_amd.parallel import DistributedDataParallel
    torch.manual_seed(0)
    m = tnn.ModuleDict({"a": tnn.Linear(4, 4), "b": tnn.Linear(4, 4)})
    ddp = DistributedDataParallel(m, bucket_mb=1e-4)
    ddp._arm()
    m["a"](torch.ones(2, 4) * (rank + 1)).sum().backward()      # "b" never gets a gradient
    ddp.finish_grad_sync()                                      # must not deadlock
    return ddp.arena.grad.clone()


def test_unused_parameters_do_not_deadlock():
    out = run_ranks(_unused_param_worker, 2)
    assert torch.equal(out[0], out[1])


# --------------------------------------------------------------------------------------------- launcher, end to end
@pytest.mark.parametrize("extra", [[], ["--sync", "ps"]])
def test_launcher_spawns_data_parallel_lenet(tmp_path, extra):
    """North-star config 1: LeNet-5, MNIST-shape synthetic data, DP on CPU/gloo world_size=2."""
    jl = os.path.join(tmp_path, "log.jsonl")
    cmd = [sys.executable, "-m", "distributed_deep_learning_amd", "--model", "lenet5", "-m", "data", "-r", "2", "-e", "2",
           "-b", "32", "--samples", "512", "--lr", "0.05", "--jsonl", jl, "--check-replicas", "2"] + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('"')]
    assert len(lines) == 2 * 3 + 1                                # rank 0 only: 3 lines per epoch + test
    assert lines[0].startswith('"train epoch 1 begins at ') and lines[-1].startswith('"test ends at ')
    recs = [json.loads(l) for l in open(jl)]
    ep = [x for x in recs if x["phase"] == "epoch"]
    assert len(ep) == 2 and ep[1]["train_loss"] < ep[0]["train_loss"]


def test_external_launcher_env_is_honoured(tmp_path):
    """torchrun-style launch: ranks come from RANK/WORLD_SIZE, not from -r."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", "-m", "distributed_deep_learning_amd", "--model", "mlp", "-e", "1", "--samples", "256",
           "--check-replicas", "1", "--global-metrics"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert sum(1 for l in r.stdout.splitlines() if l.startswith('"test ends at ')) == 1


# --------------------------------------------------------------------------------------------- failure drill
def test_killed_rank_fails_the_job_and_resume_continues(tmp_path):
    """a rank that dies mid-epoch must fail the whole job promptly (no hang), and the job must continue
    from the last checkpoint with ``--resume``."""
    import time
    ck = os.path.join(tmp_path, "ck.pt")
    res = os.path.join(tmp_path, "hist.jsonl")
    base = [sys.executable, "-m", "distributed_deep_learning_amd", "--model", "mlp", "-m", "data", "-r", "2", "-e", "3",
            "-b", "16", "--samples", "256", "--checkpoint", ck]
    t0 = time.time()
    r = subprocess.run(base + ["--inject-fault", "1:9"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0, "the job must not report success when a rank died"
    assert "[fault drill] rank 1 exits before step 9" in r.stderr
    assert time.time() - t0 < 200                                   # torn down, not waiting for a timeout
    assert os.path.exists(ck), "epoch 1 finished before the fault: its checkpoint must exist"
    r2 = subprocess.run(base + ["--resume", ck, "--jsonl", res], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-3000:]
    epochs = [json.loads(l)["epoch"] for l in open(res) if '"train_loss"' in l]
    assert epochs and epochs[0] >= 2 and epochs[-1] == 3            # continued, did not start over


# --------------------------------------------------------------------------------------------- debug ledger
def _ledger_worker(rank, world):
    from distributed_deep_learning_amd.parallel import DistributedDataParallel, ParamArena
    from distributed_deep_learning_amd.utils import GradLedger, replica_divergence
    m = _mlp(3)
    arena = ParamArena(m.named_parameters(), bucket_bytes=2048, shard_multiple=world)
    ddp = DistributedDataParallel(m, arena=arena)
    ledger = GradLedger(ddp)
    x, y = _data(rank)
    for _ in range(2):
        ((ddp(x) - y) ** 2).mean().backward()
        ddp.finish_grad_sync()
        arena.zero_grad()
    order = ledger.launch_order()
    # a bucket launched while gradients are outstanding is a contract violation the ledger must catch
    ddp._arm()                                           # noqa: SLF001
    caught = False
    try:
        ddp._launch(len(arena.buckets) - 1)              # noqa: SLF001
    except AssertionError:
        caught = True
    ddp.finish_grad_sync()
    return {"order": order, "nb": len(arena.buckets), "caught": caught, "div": replica_divergence(ddp)}


def test_grad_ledger_orders_buckets_and_catches_early_launch():
    out = run_ranks(_ledger_worker, 2)
    for o in out:
        assert o["nb"] > 1
        assert o["order"] == list(range(o["nb"] - 1, -1, -1))        # last bucket (first gradients) first
        assert o["caught"]
        assert o["div"] == 0.0


# --------------------------------------------------------------------------------------------- direct arena writers
class _DirectLinear(torch.autograd.Function):
    """y = x @ w.T whose weight gradient is written straight into ``w.grad`` (an arena view) and reported through
    ``arena.notify_grad_ready`` while autograd gets ``None`` — what the HIP wgrad / BN-backward kernels do."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x)
        ctx.w = w
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        w = ctx.w
        w.grad.add_(dy.t() @ x)
        w._ddl_arena.notify_grad_ready(w)                # noqa: SLF001
        return dy @ w.detach(), None


class _DirectMLP(tnn.Module):
    def __init__(self, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.ws = tnn.ParameterList(tnn.Parameter(torch.randn(12, 12) * 0.3) for _ in range(6))
        self.bs = tnn.ParameterList(tnn.Parameter(torch.zeros(12)) for _ in range(6))

    def forward(self, x, direct=True):
        for w, b in zip(self.ws, self.bs):
            x = torch.tanh((_DirectLinear.apply(x, w) if direct else x @ w.t()) + b)
        return x


def _direct_worker(rank, world):
    from distributed_deep_learning_amd.parallel import DistributedDataParallel, ParamArena
    from distributed_deep_learning_amd.utils import GradLedger
    m = _DirectMLP(5)
    arena = ParamArena(m.named_parameters(), bucket_bytes=8192, shard_multiple=world)     # a few tensors per bucket
    ddp = DistributedDataParallel(m, arena=arena)
    ledger = GradLedger(ddp)
    x = torch.randn(8, 12, generator=torch.Generator().manual_seed(100 + rank))
    ddp(x).pow(2).mean().backward()
    over = max(ledger.signals.values())
    ddp.finish_grad_sync()
    synced = arena.grad.clone()
    arena.zero_grad()
    local = []
    for r in range(world):                                # every rank's plain-autograd gradient, no communication
        xr = torch.randn(8, 12, generator=torch.Generator().manual_seed(100 + r))
        m(xr, direct=False).pow(2).mean().backward()
        local.append(arena.grad.clone())
        arena.zero_grad()
    return {"synced": synced, "mean": sum(local) / world, "nb": len(arena.buckets), "over": over,
            "order": ledger.launch_order()}


def test_direct_arena_writers_are_counted_once():
    """Tensors whose gradients bypass autograd's accumulation send TWO readiness signals (their own notify, then
    autograd's hook for the undefined gradient); a bucket must still wait for all of its tensors."""
    out = run_ranks(_direct_worker, 2)
    for o in out:
        assert o["nb"] > 2 and o["over"] == 2
        assert o["order"] == list(range(o["nb"] - 1, -1, -1))
        assert torch.allclose(o["synced"], o["mean"], atol=1e-6)


# --------------------------------------------------------------------------------------------- bf16 wire format
def _bf16_wire_worker(rank, world, mode):
    from distributed_deep_learning_amd.parallel import DistributedDataParallel, ParamArena
    out = {}
    for tag, dt in (("fp32", None), ("bf16", torch.bfloat16)):
        m = _mlp(5)
        arena = ParamArena(m.named_parameters(), bucket_bytes=2048, shard_multiple=world)
        ddp = DistributedDataParallel(m, mode=mode, arena=arena, grad_dtype=dt)
        x, y = _data(rank)
        ((ddp(x) - y) ** 2).mean().backward()
        ddp.finish_grad_sync()
        if mode == "ps":
            g = torch.cat([arena.grad[lo:hi] for lo, hi in ddp.owned_ranges()])
        else:
            g = arena.grad.clone()
        out[tag] = g.clone()
    return out


@pytest.mark.parametrize("mode", ["allreduce", "ps"])
def test_bf16_gradient_wire_format_tracks_fp32(mode):
    out = run_ranks(_bf16_wire_worker, 2, (mode,))
    # it really went through bf16 (a rank whose PS shards are all arena padding sees zeros either way)
    assert any(not torch.equal(o["fp32"], o["bf16"]) for o in out)
    for o in out:
        ref, got = o["fp32"], o["bf16"]
        assert torch.allclose(ref, got, rtol=2e-2, atol=2e-3 * max(ref.abs().max().item(), 1e-6))
