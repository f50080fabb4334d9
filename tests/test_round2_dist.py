"""Round-2 distributed plumbing, all on CPU / gloo: reducer watchdog, self-launching bench, ZeRO-1 (PS) mode with
sharded optimizer state and world-size-independent checkpoints, hybrid data x model parallelism, un-padded
evaluation shards, loader worker threads, device guards."""
import os
import subprocess
import sys
import threading
import time
import types

import pytest
import torch

from _dist_util import run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# --------------------------------------------------------------------------------------------- watchdog
def test_reducer_watchdog_reports_a_stuck_step_and_ignores_finished_ones():
    from distributed_deep_learning_amd.parallel.ddp import ReducerWatchdog
    fired = []
    done = threading.Event()
    wd = ReducerWatchdog(0.3, lambda tag, age: f"REPORT {tag} {age:.1f}", on_timeout=lambda tag, age: (fired.append(tag), done.set()))
    wd.note(("step", 1), lambda: True)          # finished work is retired silently
    time.sleep(0.6)
    assert not fired and not wd.fired
    wd.note(("step", 2), lambda: False)         # never finishes
    assert done.wait(5.0)
    assert fired == [("step", 2)] and wd.fired
    wd.stop()


def test_reducer_watchdog_exits_the_process_nonzero():
    """The default action: print the report and leave with exit code 3 (so a launcher tears the job down)."""
    code = ("import time, sys; sys.path.insert(0, %r)\n"
            "from distributed_deep_learning_amd.parallel.ddp import ReducerWatchdog\n"
            "wd = ReducerWatchdog(0.2, lambda tag, age: 'bucket 3 of 7 pending, launch order [6,5,4,3]')\n"
            "wd.note('s', lambda: False)\n"
            "time.sleep(10)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3
    assert "bucket 3 of 7 pending" in r.stderr


def _hang_report_worker(rank, world):
    import torch.nn as tnn
    from distributed_deep_learning_amd.parallel import DistributedDataParallel
    torch.manual_seed(0)
    m = tnn.Sequential(tnn.Linear(8, 8), tnn.Linear(8, 4))
    ddp = DistributedDataParallel(m, bucket_mb=1e-4, watchdog_s=30)
    ddp(torch.ones(2, 8)).sum().backward()
    text = ddp._hang_report((7, list(ddp._launch_order)), 31.0)       # noqa: SLF001
    ddp.finish_grad_sync()
    ddp.close()
    return text


def test_hang_report_names_buckets_pending_counts_and_launch_order():
    out = run_ranks(_hang_report_worker, 2)
    for r, text in enumerate(out):
        assert f"rank {r}/2" in text and "step 7" in text
        assert "bucket launch order" in text and "pending gradient counts" in text


# --------------------------------------------------------------------------------------------- bench launcher
def test_bench_refuses_to_report_a_smaller_job_as_n_gpus():
    """`python bench.py --gpus 2` without a launcher and without 2 GPUs: a clear error, never a 1-GPU record."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2
    assert "needs 2 GPUs" in r.stderr
    assert r.stdout.strip() == ""


def test_bench_self_launch_starts_one_child_per_gpu(monkeypatch, tmp_path):
    """The launcher half of bench.py with the GPU count faked: N children with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*
    set, a failing rank fails the job."""
    sys.path.insert(0, ROOT)
    import bench
    calls = []

    class FakeProc:
        def __init__(self, cmd, env):
            calls.append((cmd, {k: env[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
            self.rank = int(env["RANK"])

        def poll(self):
            return 5 if (self.rank == 2 and FakeProc.fail) else 0

        def terminate(self):
            pass

        def wait(self, t=None):
            return 0

        def kill(self):
            pass

    FakeProc.fail = False
    from distributed_deep_learning_amd.utils import host
    monkeypatch.setattr(host, "visible_gpus", lambda: 4)      # the launcher counts GPUs without touching HIP
    monkeypatch.setattr(bench.subprocess, "Popen", lambda cmd, env=None, stdout=None: FakeProc(cmd, env))
    a = bench.build_parser().parse_args(["--gpus", "4"])
    assert bench.self_launch(a, ["--gpus", "4"]) == 0
    assert [c[1]["RANK"] for c in calls] == ["0", "1", "2", "3"]
    assert all(c[1]["WORLD_SIZE"] == "4" and c[1]["MASTER_ADDR"] == "127.0.0.1" for c in calls)
    assert len({c[1]["MASTER_PORT"] for c in calls}) == 1
    assert all(c[0][1].endswith("bench.py") and c[0][2:] == ["--gpus", "4"] for c in calls)
    FakeProc.fail = True
    assert bench.self_launch(a, ["--gpus", "4"]) == 5


# --------------------------------------------------------------------------------------------- ZeRO-1 / PS mode
def _ps_trainer_worker(rank, world, epochs, batch, ckpt, resume):
    from distributed_deep_learning_amd.config import Config
    from distributed_deep_learning_amd.trainer import Trainer
    cfg = Config(model="mlp", mode="data", epochs=epochs, batch=batch, samples=320, sync="ps", optimizer="lars", lr=0.5,
                 weight_decay=1e-4, global_metrics=True, checkpoint=ckpt, resume=resume, bucket_mb=2e-3, graph=False)
    t = Trainer(cfg, printer=lambda *_: None).setup()
    hist = t.fit()
    opt, arena = t.optimizers[0], t.arenas[0]
    return {"loss": [h["train_loss"] for h in hist if "train_loss" in h], "epochs": [h["epoch"] for h in hist if "epoch" in h],
            "state_bytes": opt.state_bytes(), "arena_bytes": arena.total * 4, "nbuckets": len(arena.buckets),
            "w": torch.cat([p.detach().flatten() for p in t.model.parameters()])}


def test_ps_lars_checkpoint_resumes_at_a_different_world_size(tmp_path):
    """PS + LARS at world 2: optimizer state is held for 1/world of the arena; a checkpoint written after epoch 2 carries
    the momentum of EVERY shard, so resuming it at world 2 or at world 1 (same global batch) continues the
    uninterrupted run's loss curve."""
    ck = os.path.join(tmp_path, "ps.pt")
    full = run_ranks(_ps_trainer_worker, 2, (3, 16, None, None))
    assert full[0]["nbuckets"] > 1
    for r in range(2):
        assert full[r]["state_bytes"] * 2 == full[r]["arena_bytes"]      # momentum for the owned half only
    assert torch.equal(full[0]["w"], full[1]["w"])
    first = run_ranks(_ps_trainer_worker, 2, (2, 16, ck, None))
    assert first[0]["loss"] == pytest.approx(full[0]["loss"][:2], rel=1e-6)
    same = run_ranks(_ps_trainer_worker, 2, (3, 16, None, ck))
    assert same[0]["epochs"] == [3]
    assert same[0]["loss"][0] == pytest.approx(full[0]["loss"][2], rel=1e-6)
    assert torch.allclose(same[0]["w"], full[0]["w"], atol=1e-6)
    one = run_ranks(_ps_trainer_worker, 1, (3, 32, None, ck))            # different world size, same global batch
    assert one[0]["epochs"] == [3]
    assert one[0]["loss"][0] == pytest.approx(full[0]["loss"][2], rel=1e-4)
    assert torch.allclose(one[0]["w"], full[0]["w"], atol=1e-4)
    # a resume that LOST the foreign shards' momentum would not come close: check the checkpoint really holds it
    from distributed_deep_learning_amd import checkpoint as ckpt
    payload = torch.load(ck, map_location="cpu", weights_only=True)
    mom = payload["extra"]["optimizers"][0]["flat_state"]["momentum"]
    layout = payload["extra"]["optimizers"][0]["layout"]
    for name, off, n in layout:
        assert float(mom[off:off + n].abs().sum()) > 0.0, f"momentum of {name} missing from the checkpoint"
    assert ckpt.FORMAT_VERSION == payload["format"]


def _sharded_state_worker(rank, world):
    import torch.nn as tnn
    from distributed_deep_learning_amd.optim import FusedAdam
    from distributed_deep_learning_amd.parallel import DistributedDataParallel
    torch.manual_seed(1)
    m = tnn.Sequential(tnn.Linear(12, 40), tnn.Tanh(), tnn.Linear(40, 3))
    ddp = DistributedDataParallel(m, bucket_mb=2e-3, mode="ps")
    opt = FusedAdam(ddp.arena, lr=1e-2)
    ddp.attach(opt)
    g = torch.Generator().manual_seed(rank)
    for _ in range(3):
        ((ddp(torch.randn(8, 12, generator=g)) ** 2).mean()).backward()
        ddp.step(opt)
    sd = opt.state_dict()                                # collective: gathers the shards
    return {"kind": type(ddp).__name__, "bytes": opt.state_bytes(), "total": ddp.arena.total,
            "exp_avg": sd["flat_state"]["exp_avg"], "w": ddp.arena.master.clone()}


def test_ps_adam_state_is_sharded_and_gathers_to_the_full_state():
    out = run_ranks(_sharded_state_worker, 2)
    assert out[0]["kind"] == "ShardedDataParallel"                        # DistributedDataParallel(mode="ps") dispatches
    assert out[0]["bytes"] == 2 * out[0]["total"] * 4 // 2               # two Adam moments, half the arena each
    assert torch.equal(out[0]["exp_avg"], out[1]["exp_avg"])             # every rank gets the complete state
    assert out[0]["exp_avg"].numel() == out[0]["total"]
    assert torch.equal(out[0]["w"], out[1]["w"])


# --------------------------------------------------------------------------------------------- hybrid DP x MP
def _hybrid_worker(rank, world, mode, stages, batch):
    from distributed_deep_learning_amd.config import Config
    from distributed_deep_learning_amd.trainer import Trainer
    cfg = Config(model="mlp", mode=mode, stages=stages, epochs=2, batch=batch, samples=320, optimizer="sgd", lr=0.05,
                 global_metrics=True, graph=False, pipeline=4)
    t = Trainer(cfg, printer=lambda *_: None).setup()
    hist = t.fit()
    sd = t.model.state_dict()
    return {"w": torch.cat([v.detach().flatten().float() for _, v in sorted(sd.items())]),
            "loss": [h["train_loss"] for h in hist if "train_loss" in h],
            "n_reducers": sum(d is not None for d in t.ddps), "n_arenas": len(t.arenas)}


@pytest.mark.parametrize("mode", ["model", "pipeline"])
def test_hybrid_data_x_model_parallel_world2_equals_single_rank_double_batch(mode):
    """The reference's implicit hybrid (`-m model|pipeline` under an MPI launch still all-reduces across ranks,
    /root/reference/src/pytorch/CNN/main.py:134-154): 2 ranks x 2 stages, one reducer per stage.  Replicas stay identical
    and the result equals one rank training on the doubled batch (BN-free model)."""
    hyb = run_ranks(_hybrid_worker, 2, (mode, 2, 16))
    assert hyb[0]["n_reducers"] == 2 and hyb[0]["n_arenas"] == 2
    assert torch.equal(hyb[0]["w"], hyb[1]["w"])
    ref = run_ranks(_hybrid_worker, 1, ("sequential", 1, 32))
    assert torch.allclose(hyb[0]["w"], ref[0]["w"], atol=2e-5)
    assert hyb[0]["loss"] == pytest.approx(ref[0]["loss"], rel=1e-4)


# --------------------------------------------------------------------------------------------- eval shards, loader
def test_unpadded_eval_shards_cover_every_sample_exactly_once():
    from distributed_deep_learning_amd.data import ShardedSampler
    idx = torch.arange(100, 123)                                          # 23 samples over 4 ranks
    parts = [list(ShardedSampler(idx, r, 4, shuffle=False, pad=False)) for r in range(4)]
    assert sorted(sum(parts, [])) == idx.tolist()
    assert [len(p) for p in parts] == [6, 6, 6, 5]
    assert [len(ShardedSampler(idx, r, 4, shuffle=False, pad=False)) for r in range(4)] == [6, 6, 6, 5]
    padded = [list(ShardedSampler(idx, r, 4, shuffle=False)) for r in range(4)]
    assert [len(p) for p in padded] == [6, 6, 6, 6]                       # training shards still pair up


def _eval_count_worker(rank, world):
    from distributed_deep_learning_amd.config import Config
    from distributed_deep_learning_amd.trainer import Trainer
    cfg = Config(model="mlp", mode="data", epochs=1, batch=16, samples=250, global_metrics=True, graph=False)
    t = Trainer(cfg, printer=lambda *_: None).setup()
    _, _, n_val, _, _ = t._run_phase(t.val_loader, False)                 # noqa: SLF001
    _, _, n_test, _, _ = t._run_phase(t.test_loader, False)               # noqa: SLF001
    return n_val, n_test


def test_global_eval_metrics_count_each_sample_once():
    out = run_ranks(_eval_count_worker, 2)
    assert out[0] == out[1] == (25, 50)                                   # 10 % / 20 % of 250, odd split over 2 ranks


def test_loader_worker_threads_deliver_the_same_batches_in_order(tmp_path):
    from test_file_datasets import _make_pcb
    from distributed_deep_learning_amd.data import BatchLoader, ShardedSampler, files
    _make_pcb(str(tmp_path), per_class=3)
    eager = files.pcb_defects(str(tmp_path), seed=3)
    lazy = files.open_dataset("cnn", str(tmp_path), seed=3, lazy=True)
    assert isinstance(lazy, files.LazyPCBDataset) and len(lazy) == len(eager)
    for i in (0, 1, len(eager) - 1):
        assert torch.equal(lazy[i][0], eager[i][0]) and torch.equal(lazy[i][1], eager[i][1])
    idx = torch.arange(len(eager))
    base = list(BatchLoader(eager, ShardedSampler(idx, shuffle=True, seed=5), 5))
    threaded = BatchLoader(lazy, ShardedSampler(idx, shuffle=True, seed=5), 5, workers=3)
    assert threaded.workers == 3
    got = list(threaded)
    assert len(got) == len(base) == (len(eager) + 4) // 5
    for (xa, ya), (xb, yb) in zip(base, got):
        assert torch.equal(xa, xb) and torch.equal(ya, yb)
    threaded.close()


def _make_pcb6(root):
    """Six classes (the CNN has 6 outputs), two photos with one box each."""
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(1)
    for ci in range(6):
        cls = f"defect{ci}"
        os.makedirs(os.path.join(root, "images", cls))
        os.makedirs(os.path.join(root, "Annotations", cls))
        for i in range(2):
            arr = rng.integers(0, 255, size=(96, 128, 3), dtype=np.uint8)
            arr[30:70, 40:100] = 60 + ci * 30
            Image.fromarray(arr).save(os.path.join(root, "images", cls, f"p{i}.jpg"), quality=95)
            with open(os.path.join(root, "Annotations", cls, f"p{i}.xml"), "w") as f:
                f.write("<annotation><object><bndbox><xmin>38</xmin><ymin>28</ymin><xmax>98</xmax><ymax>68</ymax></bndbox>"
                        "</object></annotation>")


def test_cli_workers_flag_reaches_the_loader(tmp_path):
    from distributed_deep_learning_amd.config import parse_args
    from distributed_deep_learning_amd.trainer import Trainer
    _make_pcb6(str(tmp_path))
    cfg = parse_args(["--model", "cnn", "--data", str(tmp_path), "-w", "2", "-e", "1", "-b", "4", "--graph", "off"])
    t = Trainer(cfg, printer=lambda *_: None).setup()
    assert t.train_loader.workers == 2 and type(t.dataset).__name__ == "LazyPCBDataset"
    hist = t.fit()
    assert hist[0]["train_samples"] > 0


# --------------------------------------------------------------------------------------------- device guard
def test_kernels_refuse_operands_of_another_gpu(monkeypatch):
    """Every kernel wrapper launches on the CURRENT device's stream: an operand on another GPU must be an error, not a
    silent launch on the wrong device."""
    from distributed_deep_learning_amd.ops import _lib
    fake = types.SimpleNamespace(is_cuda=True, device=types.SimpleNamespace(index=1, type="cuda"))
    monkeypatch.setattr(_lib, "_cur_device", lambda: 0)
    with pytest.raises(_lib.WrongDeviceError):
        _lib.check_device(fake, "conv2d_fwd")
    monkeypatch.setattr(_lib, "_cur_device", lambda: 1)
    _lib.check_device(fake, "conv2d_fwd")
    switched = []
    monkeypatch.setattr(_lib, "_cur_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "set_device", lambda i: switched.append(i))
    with _lib.device_of(fake):
        pass
    assert switched == [1, 0]                                            # entered cuda:1, restored cuda:0
    with _lib.device_of(torch.zeros(1)):                                 # CPU tensors: nothing to do
        pass
    assert switched == [1, 0]
