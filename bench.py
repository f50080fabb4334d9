#!/usr/bin/env python3
"""Headline benchmark: ResNet-50 training throughput (images/sec, whole job) on synthetic 224x224 data, bf16 compute,
random-init weights — the metric/config BASELINE.json names.

    python bench.py --gpus 1 --steps 30 --warmup 10
    python bench.py --gpus 8 --steps 30 --warmup 10          # starts its own 8 ranks (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               # or under an external launcher

    python bench.py --sweep 1,2,4,8                          # one JSON line per count + scaling efficiency

One rank per GPU over RCCL (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  Without a launcher
environment ``--gpus N`` (N > 1) makes THIS process a launcher: it checks the GPU count without touching HIP (sysfs),
then starts N child ranks of this same script on 127.0.0.1 and waits for them (the reference launches its own ranks
the same way: ``torch.multiprocessing.spawn(process, nprocs=GLOBAL_WORLD)``,
``/root/reference/src/pytorch/CNN/main.py:197-203``).  Too few GPUs is an error, never a silent 1-GPU record.
``--sweep`` runs the listed counts back to back (the reference compares its modes by epoch time by hand,
``main.py:80,96``) and reports counts the machine does not have as "not measured".

With several ranks the record also says whether the replicas ended bit-identical (parameter checksums of every rank),
how long each gradient bucket's collective took on the communicator's stream, and how much of that the compute stream
actually WAITED for (exposed communication) - taken in three extra steps after the timed region, never inside it.

A step is the full training step: input conversion, forward, loss, backward with bucketed gradient all-reduce
overlapped on the communicator's stream, fused SGD-momentum update of fp32 masters + bf16 shadow.  W untimed warm-up
steps, then exactly K steps between barrier + synchronize on both sides; the reported time is the MAX over ranks.
Weak scaling: the per-GPU batch is fixed.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
# RCCL's intra-node transport needs dmabuf IPC on this host driver; must be in the environment before HIP starts.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# stock PyTorch-ROCm ResNet-50 (bench/stock_models.py) on the same MI355X, img/s per GPU, keyed by per-GPU batch
# (256 / 512: round 2; 1000: round 3, same session as our own 14.37k img/s - profiles/r3_same_session_stock_proxy_rn18.txt;
#  14.9k since the halo filter-gradient, stem and 64-channel 3x3 kernels, profiles/r3_*_kernel.txt)
STOCK_PYTORCH_1GPU_IMG_S = {256: 6638.0, 512: 6794.3, 1000: 6960.5}


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1000,
                    help="per-GPU batch size (sized for 288 GB HBM and for the conv tiles: the 256-pixel tiles of the 14x14 "
                         "layers fill the 512 resident block slots in exactly 3 rounds up to batch 1003 and need a 4th, 6 %% "
                         "full, from 1004 on; measured img/s on one MI355X: 256: 13.0k, 512: 13.7k, 668: 13.8k, 1000: 13.85k, "
                         "1024: 13.75k, 1336: 13.8k)")
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--sync", default="allreduce", choices=["allreduce", "ps"])
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "lars"])
    ap.add_argument("--bucket-mb", type=float, default=32.0)
    ap.add_argument("--image", type=int, default=224)
    ap.add_argument("--grad-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="wire format of the gradient collectives (fp32 = in place on the arena, the default)")
    ap.add_argument("--comm", default="native", choices=["torch", "native"],
                    help="collectives through the native RCCL communicator (csrc/comm.cpp, default; checked against "
                         "torch.distributed at start-up, which takes over if the check fails) or torch.distributed")
    ap.add_argument("--force-collectives", action="store_true",
                    help="diagnostic: issue the (identity) RCCL collectives of the DP reducer even on one GPU")
    ap.add_argument("--graph", default="off", choices=["on", "off"],
                    help="replay the whole step as one captured hipGraph (with several ranks: needs --comm native)")
    ap.add_argument("--sweep", default=None, metavar="N1,N2,...",
                    help="run the benchmark at each GPU count (e.g. 1,2,4,8) and print one JSON line per count plus a "
                         "summary line with the scaling efficiency; counts this machine lacks are 'not measured'")
    ap.add_argument("--comm-proxy", type=int, default=0, metavar="WORKGROUPS",
                    help="one-GPU diagnostic (with --force-collectives): behind every bucket's identity collective run a "
                         "streaming kernel of this many workgroups over the bucket on the communicator's stream - what an "
                         "RCCL ring kernel of that many channels costs the compute streams")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="self-launch only: seconds after which the launcher kills its ranks and fails")
    ap.add_argument("--watchdog-s", type=float, default=120.0,
                    help="reducer watchdog: a gradient collective pending longer than this aborts the rank with a report")
    return ap


# ----------------------------------------------------------------------------------------------- launcher
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(a, argv, capture=None) -> int:
    """``--gpus N`` with no launcher environment: start N ranks of this script.  Nothing here touches the GPU (the
    count comes from the KFD topology in sysfs; without it a throw-away child is asked), the children are fresh
    processes — never an exec of a process that already initialised HIP."""
    from distributed_deep_learning_amd.utils.host import visible_gpus
    have = visible_gpus()
    if have < a.gpus:
        print(f"bench.py: --gpus {a.gpus} needs {a.gpus} GPUs, this machine shows {have} - refusing to report a "
              f"{have}-GPU number as an {a.gpus}-GPU one", file=sys.stderr, flush=True)
        return 2
    port = _free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DDL_BENCH_SELF_LAUNCHED="1")
        out = capture if (capture is not None and r == 0) else None
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, stdout=out))
    deadline = time.monotonic() + a.launch_timeout
    rc = 0
    live = set(range(a.gpus))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py launcher: rank {r} exited with code {code}; stopping the other ranks",
                      file=sys.stderr, flush=True)
        if rc != 0 or time.monotonic() > deadline:
            if rc == 0:
                rc = 124
                print(f"bench.py launcher: ranks {sorted(live)} still running after {a.launch_timeout:.0f} s; killing them",
                      file=sys.stderr, flush=True)
            for r in live:
                procs[r].terminate()
            t_end = time.monotonic() + 10
            for r in live:
                try:
                    procs[r].wait(max(0.1, t_end - time.monotonic()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            break
        time.sleep(0.2)
    return rc


def run_sweep(a, argv) -> int:
    """One benchmark run per GPU count, each in fresh processes; efficiency = value(N) / (N * value(smallest count))."""
    import tempfile
    from distributed_deep_learning_amd.utils.host import visible_gpus
    counts = [int(v) for v in a.sweep.split(",") if v.strip()]
    have = visible_gpus()
    base_argv = []
    skip = False
    for tok in argv:                                   # everything but --sweep / --gpus and their values
        if skip:
            skip = False
            continue
        if tok in ("--sweep", "--gpus"):
            skip = True
            continue
        if tok.startswith("--sweep=") or tok.startswith("--gpus="):
            continue
        base_argv.append(tok)
    results = {}
    rc_all = 0
    for n in counts:
        if n > have:
            rec = {"n_gpus": n, "status": "not measured", "reason": f"this machine shows {have} GPU(s)"}
            print(json.dumps(rec), flush=True)
            results[n] = None
            continue
        sub = build_parser().parse_args(["--gpus", str(n), *base_argv])
        with tempfile.TemporaryFile(mode="w+") as out:
            if n == 1:
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
                rc = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", *base_argv], env=env,
                                    stdout=out).returncode
            else:
                rc = self_launch(sub, ["--gpus", str(n), *base_argv], capture=out)
            out.seek(0)
            lines = [ln for ln in out.read().splitlines() if ln.startswith("{")]
        if rc != 0 or not lines:
            rc_all = rc_all or rc or 1
            print(json.dumps({"n_gpus": n, "status": "failed", "returncode": rc}), flush=True)
            results[n] = None
            continue
        print(lines[-1], flush=True)
        results[n] = json.loads(lines[-1])
    done = [n for n in counts if results.get(n)]
    summary = {"sweep": counts, "metric": "scaling efficiency = value(N) / (N/N0 * value(N0)), weak scaling",
               "measured": done, "not_measured": [n for n in counts if not results.get(n)]}
    if done:
        n0 = done[0]
        v0 = results[n0]["value"]
        summary["baseline_gpus"] = n0
        summary["images_per_sec"] = {str(n): results[n]["value"] for n in done}
        summary["efficiency"] = {str(n): round(results[n]["value"] / (v0 * n / n0), 4) for n in done}
        summary["replicas_identical"] = {str(n): results[n].get("replicas_identical") for n in done}
    print(json.dumps(summary), flush=True)
    return rc_all


# ----------------------------------------------------------------------------------------------- one rank
class _QuietStdout:
    """Everything that libraries write to file descriptor 1 while a rank runs (gloo's connection line, RCCL's version
    banner, ...) goes to stderr: the driver contract is ONE JSON line on stdout.  ``restore()`` gives fd 1 back."""

    def __init__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def emit(self, line: str) -> None:
        """Write one line to the REAL stdout."""
        sys.stdout.flush()
        os.write(self.saved if self.saved is not None else 1, (line + "\n").encode())

    def restore(self) -> None:
        if self.saved is not None:
            sys.stdout.flush()
            os.dup2(self.saved, 1)
            os.close(self.saved)
            self.saved = None


def run_rank(a) -> int:
    quiet = _QuietStdout()
    try:
        return _run_rank(a, quiet)
    finally:
        quiet.restore()


def _run_rank(a, quiet) -> int:
    import torch
    import torch.distributed as td

    from distributed_deep_learning_amd import dist as ddist
    from distributed_deep_learning_amd import models
    from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss
    from distributed_deep_learning_amd.optim import LARS, FusedSGD
    from distributed_deep_learning_amd.parallel import DistributedDataParallel, ParamArena

    info = ddist.discover()
    if info.world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher environment says WORLD_SIZE={info.world}", file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible", file=sys.stderr)
        return 2
    if torch.cuda.device_count() <= info.local_rank:
        print(f"bench.py: rank {info.rank} wants cuda:{info.local_rank}, only {torch.cuda.device_count()} GPU(s) visible",
              file=sys.stderr)
        return 2
    dev = torch.device(f"cuda:{info.local_rank}")
    # control plane: gloo beside the native communicator (no RCCL communicator inside c10d), nccl for --comm torch
    ddist.init(info, dev, force=a.force_collectives, comm=a.comm)
    torch.manual_seed(1234)

    classes = 10 if a.model == "resnet18_cifar" else 1000
    image = 32 if a.model == "resnet18_cifar" else a.image
    model = models.build(a.model).to(dev)
    ddp = None
    comm_note = None
    if info.world > 1 or a.force_collectives:
        ddp = DistributedDataParallel(model, bucket_mb=a.bucket_mb, mode=a.sync, force_collectives=a.force_collectives,
                                      grad_dtype=torch.bfloat16 if a.grad_dtype == "bf16" else None, comm=a.comm,
                                      watchdog_s=a.watchdog_s)
        arena = ddp.arena
        comm_note = ddp.comm_note
        if a.comm_proxy > 0 and ddp.ncomm is not None and info.world == 1:
            ddp.ncomm.proxy_workgroups = a.comm_proxy
    else:
        arena = ParamArena(model.named_parameters(), bucket_bytes=int(a.bucket_mb * 2**20))
    if a.optimizer == "lars":
        opt = LARS(arena, lr=0.1, momentum=0.9, weight_decay=5e-5)
    else:
        opt = FusedSGD(arena, lr=0.01, momentum=0.9, weight_decay=1e-4)
    if ddp is not None:
        ddp.attach(opt)
    crit = SoftmaxCrossEntropyLoss()
    fwd = ddp if ddp is not None else model

    g = torch.Generator(device="cpu").manual_seed(100 + info.rank)
    x = torch.randn(a.batch, 3, image, image, generator=g).to(dev)           # random data (zeros inflate clocks)
    y = torch.randint(0, classes, (a.batch,), generator=g).to(dev)

    def step():
        loss = crit(fwd(x), y)
        loss.backward()
        if ddp is not None:
            ddp.step(opt)
        else:
            opt.step()
        return loss

    graph_on = a.graph == "on" and (ddp is None or not ddp.dist or ddp.ncomm is not None)
    if graph_on:
        from distributed_deep_learning_amd.utils import GraphedTrainStep
        graphed = GraphedTrainStep(model, crit, opt, x, y, warmup=2, ddp=ddp)

        def step():                           # noqa: F811 - same work, replayed from the captured graph
            return graphed(x, y)

    def fence():
        # every rank's GPU work is done when the barrier opens (the control plane may be a CPU-side gloo barrier), then
        # synchronize once more so nothing enqueued by the barrier itself is left
        torch.cuda.synchronize()
        if info.world > 1:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        loss = step()
    from distributed_deep_learning_amd.utils.host import freeze_gc
    freeze_gc()           # a full Python GC pass over ~270k live objects costs 70-115 ms of host time (utils/host.py)
    fence()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]     # per-step GPU time (diagnostics only)
    ms0 = torch.cuda.memory_stats(dev)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        loss = step()
        marks[i + 1].record()
    fence()
    dt_local = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    ms1 = torch.cuda.memory_stats(dev)
    alloc_diag = {k: int(ms1.get(k, 0) - ms0.get(k, 0)) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")}
    alloc_diag["reserved_gib"] = round(ms1.get("reserved_bytes.all.current", 0) / 2**30, 2)
    tmax = torch.tensor([dt_local], dtype=torch.float64)
    tmin = tmax.clone()
    if info.world > 1:
        ddist.all_reduce_any(tmax, op=td.ReduceOp.MAX)
        ddist.all_reduce_any(tmin, op=td.ReduceOp.MIN)
    dt = float(tmax.item())
    final_loss = float(loss.detach().float().item())

    # ---- after the timed region: replica check and communication diagnostics (three extra eager steps)
    replicas_identical = None
    comm_diag = None
    if ddp is not None and ddp.dist:
        cs = ddp.param_checksum()
        replicas_identical = bool(torch.equal(cs, cs[0:1].expand_as(cs)))
        if not graph_on:
            ddp.profile_sync = True
            if ddp.ncomm is not None:
                ddp.ncomm.start_profile()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            exposed = sorted(t0.elapsed_time(t1) for t0, t1 in ddp.sync_wait_events)
            per_bucket = {}
            if ddp.ncomm is not None:
                for tag, ms in ddp.ncomm.stop_profile():
                    if isinstance(tag, tuple) and tag[0] == "bucket":
                        per_bucket.setdefault(int(tag[1]), []).append(ms)
            comm_diag = {"exposed_comm_ms_per_step_median": round(exposed[len(exposed) // 2], 3) if exposed else None,
                         "bucket_collective_ms_median": {str(b): round(sorted(v)[len(v) // 2], 3)
                                                         for b, v in sorted(per_bucket.items())},
                         "note": "rank 0's view; collective time on the communicator's stream includes waiting for peers"}
            cs = ddp.param_checksum()
            replicas_identical = replicas_identical and bool(torch.equal(cs, cs[0:1].expand_as(cs)))

    if info.rank == 0:
        gb = a.batch * info.world
        value = gb * a.steps / dt
        comm_used = "none" if ddp is None or not ddp.dist else ("native" if ddp.ncomm is not None else "torch")
        rec = {
            "metric": "images/sec (whole job) ResNet-50 synthetic 224x224 training" if a.model == "resnet50"
                      else f"images/sec (whole job) {a.model} training",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": info.world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random N(0,1) images, random-init weights)",
            "config": {"model": a.model, "global_batch": gb, "per_gpu_batch": a.batch, "image": image, "seq_len": None,
                       "parallelism": f"dp{info.world}", "sync": a.sync, "optimizer": a.optimizer,
                       "bucket_mb": a.bucket_mb, "grad_wire_dtype": a.grad_dtype, "comm": comm_used, "hip_graph": graph_on},
            "rccl_ranks": (ddp.comm_ranks() if ddp is not None and ddp.dist else 1),
            "replicas_identical": replicas_identical,
            "comm_diagnostics": comm_diag,
            "control_plane": (td.get_backend() if td.is_initialized() else "none"),
            "comm_proxy_workgroups": a.comm_proxy or None,
            "ms_per_step_rank_min": round(1e3 * float(tmin.item()) / a.steps, 3),
            "ms_per_step_rank_max": round(1e3 * dt / a.steps, 3),
            "gpu_step_ms_min_median_max": [round(step_ms[0], 3), round(step_ms[len(step_ms) // 2], 3), round(step_ms[-1], 3)],
            "allocator_in_timed_region": alloc_diag,
            "bucket_plan_mib": [round((b.hi - b.lo) * 4 / 2**20, 2) for b in arena.buckets],
            "launcher": "self" if os.environ.get("DDL_BENCH_SELF_LAUNCHED") else info.launcher,
            "comm_note": comm_note,
            "vs_stock_pytorch_rocm_per_gpu": round(value / info.world / STOCK_PYTORCH_1GPU_IMG_S[a.batch], 3)
                                             if a.model == "resnet50" and a.batch in STOCK_PYTORCH_1GPU_IMG_S else None,
            "final_loss": round(final_loss, 4),
            "note": "reference publishes no numbers (BASELINE.md); comparator = stock PyTorch-ROCm measured on the same MI355X",
        }
        quiet.emit(json.dumps(rec))
    if ddp is not None:
        ddp.close()
    ddist.shutdown()
    return 0


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else list(argv)
    a = build_parser().parse_args(argv)
    if a.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    from distributed_deep_learning_amd import dist as ddist
    info = ddist.discover()
    if a.sweep:
        if info.launcher != "none":
            print("bench.py: --sweep starts its own ranks; run it without a launcher", file=sys.stderr)
            return 2
        return run_sweep(a, argv)
    if info.world == 1 and a.gpus > 1 and info.launcher == "none":
        return self_launch(a, argv)
    return run_rank(a)


if __name__ == "__main__":
    sys.exit(main())
