"""Per-rank setup and the train / validate / test loop — each exactly once for every workload (the reference
copy-pastes ``getConfiguration / process / worker / main`` per workload: ``/root/reference/src/pytorch/CNN/main.py``
vs ``LSTM/main.py`` vs the flat ``MLP/main.py``).

``Trainer.setup()`` is the reference's ``process()`` (``CNN/main.py:129-184``): seed, devices, model, parallel
wrapper, loss, optimizer, scheduler, split + sharded samplers + loaders, optional resume.  ``Trainer.fit()`` is its
``worker()`` (``:76-127``): epochs x (train, validate) then test, printing the same four log lines from rank 0.
Differences are the fixes listed in SURVEY Appendix A: gradients really are synchronised in every multi-rank mode
(A1/A2), parameters are broadcast from rank 0 (A6), splits are disjoint and honoured (A3), ranks bind their own GPU
(A18), no per-step host sync.
"""
from __future__ import annotations

import os
import time
from typing import List, Optional

import torch
import torch.nn as tnn

from . import checkpoint as ckpt
from . import dist as ddist
from . import models
from .config import Config
from .data import BatchLoader, ShardedSampler, WindowDataset, split_indices
from .data import synthetic as syn
from .metrics import (JsonlLogger, PhaseMeter, StepTimer, fmt_test_end, fmt_train_begin, fmt_train_end, fmt_val_end,
                      now_ts)
from .nn import L1Loss, SoftmaxCrossEntropyLoss
from .optim import LARS, FusedAdam, FusedSGD, StepLR
from .parallel import DistributedDataParallel, ParamArena, PipelinedModel, StagedModel

WORKLOADS = {
    # name           dataset factory                     loss    optimizer  model kwargs from (-l, -s)
    "cnn":          (lambda n, s: syn.pcb_like(n, s),            "xent", "sgd"),
    "densenet_ref": (lambda n, s: syn.pcb_like(n, s),            "xent", "sgd"),
    "lstm":         (lambda n, s: WindowDataset(seed=s),         "l1",   "adam"),
    "lstm_ref":     (lambda n, s: WindowDataset(seed=s),         "l1",   "adam"),
    "mlp":          (lambda n, s: syn.tabular_dataset(n, seed=s), "xent", "adam"),
    "mlp_ref":      (lambda n, s: syn.tabular_dataset(n, seed=s), "xent", "adam"),
    "lenet5":       (lambda n, s: syn.mnist_like(n, s),          "xent", "sgd"),
    "resnet18_cifar": (lambda n, s: syn.cifar_like(n, s),        "xent", "sgd"),
}


def _model_kwargs(cfg: Config) -> dict:
    kw = {}
    m = cfg.model
    if m in ("cnn", "densenet_ref"):
        if cfg.nlayers is not None:
            kw["dense_layers"] = cfg.nlayers
        if cfg.size is not None:
            kw["bn_size"] = cfg.size
    elif m in ("lstm", "lstm_ref"):
        if cfg.nlayers is not None:
            kw["hidden_layers"] = cfg.nlayers
        if cfg.size is not None:
            kw["hidden_params"] = cfg.size
    elif m in ("mlp", "mlp_ref"):
        if cfg.nlayers is not None:
            kw["hidden_layers"] = cfg.nlayers
    if cfg.classes is not None and m not in ("lstm", "lstm_ref", "cnn", "densenet_ref", "mlp", "mlp_ref"):
        kw["num_classes"] = cfg.classes
    return kw


# Workloads whose step is launch-bound (hundreds of microsecond-kernels): with ``--graph auto`` (the default) a
# single-process GPU run replays them from one captured hipGraph (2.7x-5.4x the eager step at batch 32,
# profiles/r1_reference_workloads_vs_stock.txt).  Everything else runs eagerly unless ``--graph on``.
_GRAPH_AUTO_MODELS = frozenset({"cnn", "densenet_ref", "lstm", "lstm_ref", "mlp", "mlp_ref", "lenet5", "resnet18_cifar"})


class Trainer:
    def __init__(self, cfg: Config, info: Optional[ddist.RankInfo] = None, verbose: Optional[bool] = None,
                 printer=print):
        self.cfg = cfg.validate()
        self.info = info or ddist.discover()
        self.verbose = (self.info.rank == 0) if verbose is None else verbose
        self.print = printer
        self.epoch = 0
        self.global_step = 0
        self.history: List[dict] = []

    # ------------------------------------------------------------------ setup (reference: process())
    def _devices(self) -> List[torch.device]:
        cfg, info = self.cfg, self.info
        if cfg.device == "cpu":
            n = cfg.stages if cfg.mode in ("model", "pipeline") else 1
            return [torch.device("cpu")] * n
        if not torch.cuda.is_available():
            raise RuntimeError("device 'gpu' requested but no GPU is visible")
        if cfg.mode in ("model", "pipeline"):
            base, count = info.local_rank * cfg.stages, torch.cuda.device_count()
            if base + cfg.stages > count:
                # fewer GPUs than stages: stages share devices round-robin (the reference's LSTM self-test runs its four
                # "devices" on one CPU the same way, LSTM/model.py:180-192) - exercises the staging code on a 1-GPU box
                if info.rank == 0:
                    print(f"[trainer] {cfg.mode} mode: {cfg.stages} stages on {count} visible GPU(s), sharing devices",
                          flush=True)
                return [torch.device(f"cuda:{(base + i) % count}") for i in range(cfg.stages)]
            return [torch.device(f"cuda:{base + i}") for i in range(cfg.stages)]
        return [torch.device(f"cuda:{info.local_rank}")]

    def setup(self) -> "Trainer":
        cfg, info = self.cfg, self.info
        self.devices = self._devices()
        self.in_device, self.out_device = self.devices[0], self.devices[-1]
        external = info.launcher in ("torchrun", "mpi", "slurm") and "MASTER_PORT" in os.environ
        # control plane: gloo beside the native RCCL data plane (one communicator per stage device, none of them bound
        # to the process group); ``--comm torch``: an nccl group, left UNBOUND when this rank drives several GPUs
        ddist.init(info, self.in_device, master_addr=None if external else cfg.master_addr,
                   master_port=None if external else cfg.master_port, comm=cfg.comm,
                   bind_device=cfg.mode not in ("model", "pipeline"))
        torch.manual_seed(cfg.seed)                           # reference: torch.manual_seed(42), main.py:132
        from .ops import kernels as _kernels
        from .ops import state as _ops_state
        _kernels.seed_dropout(info.rank)                      # same weights everywhere, but every replica its own masks
        if cfg.deterministic:
            _ops_state.set_deterministic(True)

        model = models.build(cfg.model, **_model_kwargs(cfg))
        if cfg.mode == "pipeline":
            model = PipelinedModel(model, self.devices, micro_batch=cfg.pipeline)
        elif cfg.mode == "model":
            model = StagedModel(model, self.devices)
        else:
            model = model.to(self.in_device)
        self.model = model

        # one arena (+ reducer when world > 1) per device
        groups = list(model.stages) if isinstance(model, StagedModel) else [model]
        self.arenas: List[ParamArena] = []
        self.ddps: List[Optional[DistributedDataParallel]] = []
        for gi, g in enumerate(groups):
            named = [(f"s{gi}.{n}", p) for n, p in g.named_parameters()]
            if not named:
                self.ddps.append(None)
                continue
            if info.world > 1:
                # the arena is built here, with the same stage-qualified names as the single-rank path, so optimizer
                # state in a checkpoint maps onto any world size / sync mode by name
                arena = ParamArena(named, bucket_bytes=int(cfg.bucket_mb * 2**20),
                                   shard_multiple=info.world if cfg.sync == "ps" else 1)
                d = DistributedDataParallel(g, bucket_mb=cfg.bucket_mb, mode=cfg.sync, arena=arena,
                                            grad_dtype=torch.bfloat16 if cfg.grad_dtype == "bf16" else None,
                                            comm=cfg.comm if self.in_device.type == "cuda" else "torch")
                if isinstance(model, StagedModel):
                    model.stages[gi] = d
                else:
                    self.model = d
                self.ddps.append(d)
                self.arenas.append(d.arena)
            else:
                self.ddps.append(None)
                self.arenas.append(ParamArena(named, bucket_bytes=int(cfg.bucket_mb * 2**20)))

        wl = WORKLOADS.get(cfg.model)
        loss_kind = cfg.loss or (wl[1] if wl else "xent")
        self.criterion = SoftmaxCrossEntropyLoss() if loss_kind == "xent" else L1Loss()
        opt_kind = cfg.optimizer if (cfg.optimizer != "sgd" or wl is None) else wl[2]
        self.optimizers = []
        for a in self.arenas:
            if opt_kind == "adam":
                o = FusedAdam(a, lr=cfg.lr if cfg.lr != 0.01 else 1e-3, weight_decay=cfg.weight_decay)
            elif opt_kind == "lars":
                o = LARS(a, lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
            else:
                o = FusedSGD(a, lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
            self.optimizers.append(o)
        live = [d for d in self.ddps if d is not None]
        for d, o in zip(live, self.optimizers):
            d.attach(o)
        # reference: StepLR(step_size=7, gamma=0.1) for the CNN only (main.py:161); harmless elsewhere
        self.schedulers = [StepLR(o, step_size=cfg.lr_step, gamma=cfg.lr_gamma) for o in self.optimizers]

        self._build_data(wl)
        self.jsonl = JsonlLogger(cfg.jsonl, enabled=info.rank == 0)
        if cfg.resume:
            self.load(cfg.resume)
        return self

    def _build_data(self, wl) -> None:
        cfg, info = self.cfg, self.info
        if cfg.data:
            from .data.files import open_dataset
            ds = open_dataset(cfg.model, cfg.data, seed=cfg.seed, lazy=cfg.nworkers > 0)
        elif wl is not None:
            ds = wl[0](cfg.samples, cfg.seed)
        elif cfg.model.startswith("vgg") or cfg.model in ("resnet18", "resnet34", "resnet50", "resnet101"):
            ds = syn.imagenet_like(cfg.samples, cfg.seed, classes=cfg.classes or 1000)
        else:
            raise KeyError(f"no default dataset for model {cfg.model}")
        if hasattr(ds, "to") and self.in_device.type == "cuda" and not (cfg.data and cfg.nworkers > 0):
            ds = ds.to(self.in_device)                       # synthetic data lives on the device: zero H2D per step
        self.dataset = ds
        tr, va, te = split_indices(len(ds), cfg.split, seed=cfg.seed)
        # training shards are padded to equal length (collectives must pair up); evaluation shards are NOT, so
        # metrics reduced over ranks count every sample exactly once
        mk = lambda idx, train: BatchLoader(  # noqa: E731
            ds, ShardedSampler(idx, info.rank, info.world, shuffle=train, seed=cfg.seed, pad=train), cfg.batch,
            device=self.in_device, prefetch=cfg.nworkers > 0, workers=cfg.nworkers)
        self.train_loader, self.val_loader, self.test_loader = mk(tr, True), mk(va, False), mk(te, False)

    # ------------------------------------------------------------------ one step
    def _forward_loss(self, x, y):
        out = self.model(x)
        y = y.to(out.device, non_blocking=True)
        loss = self.criterion(out, y)
        return loss, self.criterion.last_correct

    def _graph_step(self, x, y):
        """The captured-graph path (``--graph``): single process, one device, fixed batch shape."""
        cfg = self.cfg
        want = cfg.graph if cfg.graph is not None else cfg.model in _GRAPH_AUTO_MODELS
        # several ranks: whenever the collectives can be captured, i.e. the native communicator moves the gradients
        live = [d for d in getattr(self, "ddps", []) if d is not None]
        multi_ok = cfg.comm == "native" and len(live) == 1 and live[0].ncomm is not None
        ok = (want and (self.info.world == 1 or multi_ok) and cfg.mode in ("sequential", "data") and x.is_cuda
              and len(self.optimizers) == 1 and hasattr(self.criterion, "last_correct"))
        if not ok:
            return None
        g = getattr(self, "_graphed", None)
        if g is None:
            if x.shape[0] != cfg.batch:
                return None
            from .utils.graph import GraphedTrainStep
            y = y.to(x.device)
            ddp = self.ddps[0] if self.ddps and self.ddps[0] is not None else None
            model = ddp.module if ddp is not None else self.model
            g = self._graphed = GraphedTrainStep(model, self.criterion, self.optimizers[0], x, y, warmup=1, ddp=ddp)
            self.global_step += 1                      # the warm-up step trained on this batch
            return g.warm_loss, g.warm_correct
        if x.shape != g.x.shape:
            return None                                # ragged last batch: eager
        loss = g(x, y.to(x.device))
        self.global_step += 1
        return loss, g.correct

    def _maybe_die(self) -> None:
        """Fault drill (``--inject-fault RANK:STEP``): simulate a killed rank.  The launcher must notice (spawn tears
        the job down / the survivors' collectives time out) and the run must be resumable from the last checkpoint."""
        spec = self.cfg.inject_fault
        if spec:
            r, st = (int(v) for v in spec.split(":"))
            if r == self.info.rank and self.global_step + 1 == st:
                import os
                import sys
                print(f"[fault drill] rank {r} exits before step {st}", file=sys.stderr, flush=True)
                os._exit(17)

    def _graph_eval(self, x, y):
        """Evaluation batch through a captured forward graph (same switch as the training graph; one graph per batch
        shape, so the ragged last batch of a split gets its own or runs eagerly)."""
        cfg = self.cfg
        want = cfg.graph if cfg.graph is not None else cfg.model in _GRAPH_AUTO_MODELS
        if not (want and cfg.mode in ("sequential", "data") and x.is_cuda and len(self.optimizers) == 1
                and hasattr(self.criterion, "last_correct")):
            return None
        cache = self.__dict__.setdefault("_graphed_eval", {})
        key = (tuple(x.shape), tuple(y.shape), x.dtype, y.dtype)
        g = cache.get(key)
        if g is None:
            if len(cache) >= 4 or x.shape[0] != cfg.batch:
                return None
            from .utils.graph import GraphedEvalStep
            model = self.model.module if isinstance(self.model, DistributedDataParallel) else self.model
            g = cache[key] = GraphedEvalStep(model, self.criterion, x, y.to(x.device))
        return g(x, y.to(x.device))

    def train_step(self, x, y):
        self._maybe_die()
        out = self._graph_step(x, y)
        if out is not None:
            return out
        if isinstance(self.model, PipelinedModel) and self.cfg.pipeline_schedule == "1f1b":
            # every micro-batch's backward is issued as it leaves the last stage (parallel/pipeline.py)
            loss, correct = self.model.train_step(x, y, self.criterion, reducers=self.ddps)
        else:
            loss, correct = self._forward_loss(x, y)
            loss.backward()
        di = 0
        for d in self.ddps:
            if d is None:
                continue
            d.step(self.optimizers[di])
            di += 1
        if di == 0:
            for o in self.optimizers:
                o.step()
        self.global_step += 1
        if self.cfg.check_replicas and self.global_step % self.cfg.check_replicas == 0:
            self.assert_replicas_equal()
        return loss, correct

    def assert_replicas_equal(self) -> None:
        for d in self.ddps:
            if d is None:
                continue
            cs = d.param_checksum()
            if not torch.equal(cs, cs[0:1].expand_as(cs)):
                raise RuntimeError(f"replica divergence detected at step {self.global_step}: {cs.tolist()}")

    # ------------------------------------------------------------------ phases (reference: worker())
    def _run_phase(self, loader, train: bool):
        meter = PhaseMeter(self.out_device)
        timer = StepTimer(self.out_device) if train else None
        self.model.train(train)
        ctx = torch.enable_grad() if train else torch.no_grad()
        with ctx:
            for i, (x, y) in enumerate(loader):
                if self.cfg.max_steps is not None and i >= self.cfg.max_steps:
                    break
                if train:
                    loss, correct = self.train_step(x, y)
                    timer.tick()
                else:
                    out = self._graph_eval(x, y)
                    loss, correct = out if out is not None else self._forward_loss(x, y)
                meter.update(loss, correct, len(x))
        acc, loss, n, dt = meter.result(reduce=self.cfg.global_metrics)
        return acc, loss, n, dt, (timer.summary() if timer else {})

    def fit(self) -> List[dict]:
        cfg = self.cfg
        from .utils.host import freeze_gc, unfreeze_gc
        freeze_gc()             # model, arena and loaders exist: keep full GC passes off the launch-bound step loop
        try:
            return self._fit(cfg)
        finally:
            unfreeze_gc()

    def _fit(self, cfg) -> List[dict]:
        for epoch in range(self.epoch + 1, cfg.epochs + 1):
            self.train_loader.set_epoch(epoch)
            if self.verbose:
                self.print(fmt_train_begin(epoch, now_ts()))
            acc, loss, n, dt, st = self._run_phase(self.train_loader, True)
            if self.verbose:
                self.print(fmt_train_end(epoch, now_ts(), acc, loss))
            rec = {"epoch": epoch, "train_acc": acc, "train_loss": loss, "train_samples": n, "train_s": dt,
                   "samples_per_s": n * (self.info.world if not cfg.global_metrics else 1) / max(dt, 1e-9), **st}
            vacc, vloss, vn, vdt, _ = self._run_phase(self.val_loader, False)
            if self.verbose:
                self.print(fmt_val_end(epoch, now_ts(), vacc, vloss))
            rec.update(val_acc=vacc, val_loss=vloss)
            for s in self.schedulers:
                s.step()                                       # reference: lrDecay.step() per epoch, main.py:112
            self.epoch = epoch
            self.history.append(rec)
            self.jsonl.log(phase="epoch", **rec)
            if cfg.checkpoint:
                self.save(cfg.checkpoint)
        tacc, tloss, tn, tdt, _ = self._run_phase(self.test_loader, False)
        if self.verbose:
            self.print(fmt_test_end(now_ts(), tacc, tloss))
        self.history.append({"test_acc": tacc, "test_loss": tloss})
        self.jsonl.log(phase="test", test_acc=tacc, test_loss=tloss)
        ddist.barrier()                                        # reference: final barrier, main.py:183-184
        return self.history

    # ------------------------------------------------------------------ checkpoint
    def _ckpt_model(self):
        return self.model

    def save(self, path: str) -> None:
        """Collective: every rank calls it (PS mode gathers master and optimizer-state shards), rank 0 writes."""
        for d in self.ddps:
            if d is not None:
                d.prepare_checkpoint()
        opt_state = [o.state_dict() for o in self.optimizers]
        sch_state = [s.state_dict() for s in self.schedulers]
        ckpt.save(path, self._ckpt_model(), epoch=self.epoch, step=self.global_step,
                  sampler=self.train_loader.sampler, extra={"optimizers": opt_state, "schedulers": sch_state,
                                                            "config": self.cfg.to_json()})

    def load(self, path: str) -> None:
        meta = ckpt.load(path, self._ckpt_model(), sampler=self.train_loader.sampler)
        for a in self.arenas:
            a.sync_shadow()
        for d in self.ddps:
            if d is not None:
                d.sync_parameters()
        for o, sd in zip(self.optimizers, meta["extra"].get("optimizers", [])):
            o.load_state_dict(sd)
        for s, sd in zip(self.schedulers, meta["extra"].get("schedulers", [])):
            s.load_state_dict(sd)
        self.epoch, self.global_step = meta["epoch"], meta["step"]


def run(cfg: Config, info: Optional[ddist.RankInfo] = None, printer=print) -> List[dict]:
    t = Trainer(cfg, info, printer=printer).setup()
    return t.fit()
