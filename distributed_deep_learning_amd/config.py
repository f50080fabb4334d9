"""One typed configuration for every workload and run mode.

Keeps the reference's CLI (``-l -s -e -b -d -w -m -p -r``, ``/root/reference/src/pytorch/CNN/main.py:47-59``) and adds
what it hard-codes (lr, momentum, seed, split, model choice, sync mode, bucket size, checkpointing).  The
reference's derived keys ``GLOBAL_RANK / LOCAL_RANK / GLOBAL_WORLD / LOCAL_WORLD / DISTRIBUTED`` become
:class:`..dist.RankInfo`, filled from the launcher environment instead of a scan for ``'MPI_'`` in ``os.environ``.
"""
from __future__ import annotations

import argparse
import json
from dataclasses import asdict, dataclass, field
from typing import List, Optional, Sequence

MODES = ("sequential", "model", "pipeline", "data")
SYNC_MODES = ("allreduce", "ps")
OPTIMIZERS = ("sgd", "lars", "adam")


@dataclass
class Config:
    # --- reference flags ---
    model: str = "cnn"                 # workload / architecture (models.REGISTRY key)
    nlayers: Optional[int] = None      # -l  CNN: layers per dense block; LSTM / MLP: hidden layers
    size: Optional[int] = None         # -s  CNN: bn_size; LSTM: hidden size
    epochs: int = 10                   # -e
    batch: int = 32                    # -b  per-rank batch size
    device: str = "cpu"                # -d  cpu | gpu
    nworkers: int = 0                  # -w  loader threads assembling pinned host batches ahead (file-backed data)
    mode: str = "sequential"           # -m  sequential | model | pipeline | data
    pipeline: int = 2                  # -p  micro-batch size
    pipeline_schedule: str = "1f1b"    # training schedule of -m pipeline: "1f1b" (a micro-batch's backward is issued as
                                       # soon as it leaves the last stage) | "gpipe" (the reference: all forwards, then
                                       # one backward over everything)
    run: int = 1                       # -r  local ranks to spawn for -m data
    # --- made explicit ---
    stages: int = 2                    # devices for model / pipeline mode
    lr: float = 0.01
    momentum: float = 0.9
    weight_decay: float = 0.0
    optimizer: str = "sgd"
    lr_step: int = 7
    lr_gamma: float = 0.1
    seed: int = 42
    split: Sequence[float] = (0.7, 0.1, 0.2)
    samples: int = 2048                # synthetic dataset size
    data: Optional[str] = None         # dataset path (PCB folder / CSV); default = synthetic data of the same shape
    classes: Optional[int] = None
    sync: str = "allreduce"            # gradient sync: all-reduce | ps (reduce-scatter / all-gather)
    bucket_mb: float = 32.0
    grad_dtype: str = "fp32"           # wire format of the gradient collectives: fp32 (in place) | bf16
    comm: str = "native"               # GPU collectives: "native" (direct RCCL, csrc/comm.cpp) | "torch" (torch.distributed)
    loss: Optional[str] = None         # xent | l1 (default per workload)
    checkpoint: Optional[str] = None   # path to save to after every epoch
    resume: Optional[str] = None
    jsonl: Optional[str] = None
    global_metrics: bool = False       # all-reduce metrics over ranks (reference reports rank 0's shard)
    graph: Optional[bool] = None       # hipGraph replay of the train step: True / False / None = auto (on for the
                                       # launch-bound workloads: cnn, lstm, mlp, lenet5, resnet18_cifar; 1 process, GPU)
    deterministic: bool = False        # bit-reproducible weight gradients (slabs + ordered reduce instead of float
                                       # atomics, csrc/conv_wgrad.hip; 0.6-1.3 % slower on ResNet-50)
    check_replicas: int = 0            # every k steps compare parameter checksums across ranks (debug)
    max_steps: Optional[int] = None    # stop each epoch early (smoke tests)
    inject_fault: Optional[str] = None  # "RANK:STEP": that rank dies (hard exit) before that global step - fault drill
    master_addr: str = "127.0.0.1"
    master_port: int = 29500

    def validate(self) -> "Config":
        if self.mode not in MODES:
            raise ValueError(f"mode must be one of {MODES}")
        if self.device not in ("cpu", "gpu"):
            raise ValueError("device must be cpu or gpu")
        if self.sync not in SYNC_MODES:
            raise ValueError(f"sync must be one of {SYNC_MODES}")
        if self.optimizer not in OPTIMIZERS:
            raise ValueError(f"optimizer must be one of {OPTIMIZERS}")
        if self.pipeline_schedule not in ("1f1b", "gpipe"):
            raise ValueError("pipeline_schedule must be 1f1b or gpipe")
        if self.batch < 1 or self.epochs < 0 or self.run < 1 or self.pipeline < 1 or self.stages < 1:
            raise ValueError("batch, run, pipeline and stages must be >= 1")
        return self

    def to_json(self) -> str:
        d = asdict(self)
        d["split"] = list(self.split)
        return json.dumps(d)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="ddl-amd", description="MI355X-native distributed training")
    p.add_argument("--model", default="cnn")
    p.add_argument("-l", "--nlayers", type=int, nargs="?", default=None)
    p.add_argument("-s", "--size", type=int, nargs="?", default=None)
    p.add_argument("-e", "--epochs", type=int, nargs="?", default=10)
    p.add_argument("-b", "--batch", type=int, nargs="?", default=32)
    p.add_argument("-d", "--device", choices=["cpu", "gpu"], nargs="?", default="cpu")
    p.add_argument("-w", "--nworkers", type=int, nargs="?", default=0)
    p.add_argument("-m", "--mode", choices=list(MODES), nargs="?", default="sequential")
    p.add_argument("-p", "--pipeline", type=int, nargs="?", default=2)
    p.add_argument("-r", "--run", type=int, nargs="?", default=1)
    p.add_argument("--pipeline-schedule", dest="pipeline_schedule", choices=["1f1b", "gpipe"], default="1f1b")
    p.add_argument("--stages", type=int, default=2)
    p.add_argument("--lr", type=float, default=0.01)
    p.add_argument("--momentum", type=float, default=0.9)
    p.add_argument("--weight-decay", dest="weight_decay", type=float, default=0.0)
    p.add_argument("--optimizer", choices=list(OPTIMIZERS), default="sgd")
    p.add_argument("--lr-step", dest="lr_step", type=int, default=7)
    p.add_argument("--lr-gamma", dest="lr_gamma", type=float, default=0.1)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--samples", type=int, default=2048)
    p.add_argument("--data", default=None, help="dataset root (cnn: PCB folder) or CSV (lstm, mlp); default synthetic")
    p.add_argument("--classes", type=int, default=None)
    p.add_argument("--sync", choices=list(SYNC_MODES), default="allreduce")
    p.add_argument("--bucket-mb", dest="bucket_mb", type=float, default=32.0)
    p.add_argument("--grad-dtype", dest="grad_dtype", choices=["fp32", "bf16"], default="fp32")
    p.add_argument("--comm", choices=["torch", "native"], default="native",
                   help="GPU gradient collectives: native RCCL communicator (default) or torch.distributed; CPU runs use gloo")
    p.add_argument("--loss", choices=["xent", "l1"], default=None)
    p.add_argument("--checkpoint", default=None)
    p.add_argument("--resume", default=None)
    p.add_argument("--jsonl", default=None)
    p.add_argument("--global-metrics", dest="global_metrics", action="store_true")
    p.add_argument("--graph", nargs="?", const="on", default="auto", choices=["on", "off", "auto"],
                   help="replay the training step as one captured hipGraph (auto: on for the small, launch-bound workloads)")
    p.add_argument("--deterministic", action="store_true",
                   help="bit-reproducible weight gradients (ordered reduce of the split-M partial tiles instead of atomics)")
    p.add_argument("--check-replicas", dest="check_replicas", type=int, default=0)
    p.add_argument("--max-steps", dest="max_steps", type=int, default=None)
    p.add_argument("--inject-fault", dest="inject_fault", default=None, metavar="RANK:STEP",
                   help="fault drill: the given rank exits hard before the given global step")
    p.add_argument("--master-addr", dest="master_addr", default="127.0.0.1")
    p.add_argument("--master-port", dest="master_port", type=int, default=29500)
    return p


def parse_args(argv: Optional[List[str]] = None) -> Config:
    ns = build_parser().parse_args(argv)
    ns.graph = {"on": True, "off": False, "auto": None}[ns.graph]
    return Config(**vars(ns)).validate()
