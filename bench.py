#!/usr/bin/env python3
"""Headline benchmark: ResNet-50 training throughput (images/sec, whole job) on synthetic 224x224 data, bf16 compute,
random-init weights — the metric/config BASELINE.json names.

    python bench.py --gpus 1 --steps 30 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One rank per GPU over RCCL (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  A step is the full
training step: input conversion, forward, loss, backward with bucketed gradient all-reduce overlapped on RCCL's
stream, fused SGD-momentum update of fp32 masters + bf16 shadow.  W untimed warm-up steps, then exactly K steps
between barrier + synchronize on both sides; the reported time is the MAX over ranks.  Weak scaling: the per-GPU
batch is fixed.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
# RCCL's intra-node transport needs dmabuf IPC on this host driver; must be in the environment before HIP starts.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# stock PyTorch-ROCm ResNet-50 (bench/stock_models.py) on the same MI355X, img/s per GPU, keyed by per-GPU batch
STOCK_PYTORCH_1GPU_IMG_S = {256: 6638.0, 512: 6793.5}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024,
                    help="per-GPU batch size (sized for 288 GB HBM; measured img/s on one MI355X: 256: 12.7k, 512: 13.4k, "
                         "640: 13.7k, 1024: 13.8k)")
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--sync", default="allreduce", choices=["allreduce", "ps"])
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "lars"])
    ap.add_argument("--bucket-mb", type=float, default=32.0)
    ap.add_argument("--image", type=int, default=224)
    ap.add_argument("--grad-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="wire format of the gradient collectives (fp32 = in place on the arena, the default)")
    ap.add_argument("--comm", default="torch", choices=["torch", "native"],
                    help="collectives through torch.distributed (default) or the native RCCL communicator (csrc/comm.cpp)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="diagnostic: issue the (identity) RCCL collectives of the DP reducer even on one GPU")
    ap.add_argument("--graph", default="off", choices=["on", "off"],
                    help="replay the whole step as one captured hipGraph (with several ranks: needs --comm native)")
    a = ap.parse_args()

    import torch
    import torch.distributed as td

    from distributed_deep_learning_amd import dist as ddist
    from distributed_deep_learning_amd import models
    from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss
    from distributed_deep_learning_amd.optim import LARS, FusedSGD
    from distributed_deep_learning_amd.parallel import DistributedDataParallel, ParamArena

    info = ddist.discover()
    if info.world != a.gpus:
        if info.world == 1 and a.gpus > 1:
            print(f"bench.py: --gpus {a.gpus} needs a launcher (torch.distributed.run); running on 1 GPU", file=sys.stderr)
        a.gpus = info.world
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible", file=sys.stderr)
        return 2
    dev = torch.device(f"cuda:{info.local_rank}")
    ddist.init(info, dev, force=a.force_collectives)
    torch.manual_seed(1234)

    classes = 10 if a.model == "resnet18_cifar" else 1000
    image = 32 if a.model == "resnet18_cifar" else a.image
    model = models.build(a.model).to(dev)
    ddp = None
    if info.world > 1 or a.force_collectives:
        ddp = DistributedDataParallel(model, bucket_mb=a.bucket_mb, mode=a.sync, force_collectives=a.force_collectives,
                                      grad_dtype=torch.bfloat16 if a.grad_dtype == "bf16" else None, comm=a.comm)
        arena = ddp.arena
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

    graph_on = a.graph == "on" and (ddp is None or not ddp.dist or a.comm == "native")
    if graph_on:
        from distributed_deep_learning_amd.utils import GraphedTrainStep
        graphed = GraphedTrainStep(model, crit, opt, x, y, warmup=2, ddp=ddp)

        def step():                           # noqa: F811 - same work, replayed from the captured graph
            return graphed(x, y)

    def fence():
        if info.world > 1:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        loss = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if info.world > 1:
        td.all_reduce(t, op=td.ReduceOp.MAX)
    dt = float(t.item())
    final_loss = float(loss.detach().float().item())

    if info.rank == 0:
        gb = a.batch * info.world
        value = gb * a.steps / dt
        rec = {
            "metric": "images/sec (whole job) ResNet-50 synthetic 224x224 training" if a.model == "resnet50"
                      else f"images/sec (whole job) {a.model} training",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": info.world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random N(0,1) images, random-init weights)",
            "config": {"model": a.model, "global_batch": gb, "per_gpu_batch": a.batch, "image": image, "seq_len": None,
                       "parallelism": f"dp{info.world}", "sync": a.sync, "optimizer": a.optimizer,
                       "bucket_mb": a.bucket_mb, "grad_wire_dtype": a.grad_dtype, "comm": a.comm, "hip_graph": graph_on},
            "vs_stock_pytorch_rocm_per_gpu": round(value / info.world / STOCK_PYTORCH_1GPU_IMG_S[a.batch], 3)
                                             if a.model == "resnet50" and a.batch in STOCK_PYTORCH_1GPU_IMG_S else None,
            "final_loss": round(final_loss, 4),
            "note": "reference publishes no numbers (BASELINE.md); comparator = stock PyTorch-ROCm measured on the same MI355X",
        }
        print(json.dumps(rec), flush=True)
    ddist.shutdown()
    return 0


if __name__ == "__main__":
    sys.exit(main())
