#!/usr/bin/env python3
"""The reference's three workloads (CNN = DenseNet-BC variant on 64x64 crops, LSTM, MLP;
/root/reference/src/pytorch/{CNN,LSTM,MLP}/model.py) as full training steps on one MI355X:

    ours, eager     this framework's kernels, step launched from Python
    ours, hipGraph  the same step replayed from one captured graph (utils.GraphedTrainStep)
    stock fp32      the same architecture from plain torch.nn modules in fp32 (what the reference runs), eager
    stock bf16      the same under torch.autocast(bfloat16) (+ channels_last for the CNN): the SAME-PRECISION comparator
                    for our bf16 CNN / LSTM paths (our MLP path is fp32, so its comparator is the fp32 column)

    python bench/ref_workloads.py [--batches 32,512] [--steps 100]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as tnn

from distributed_deep_learning_amd import models
from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss
from distributed_deep_learning_amd.optim import FusedAdam, FusedSGD
from distributed_deep_learning_amd.parallel import ParamArena
from distributed_deep_learning_amd.utils import GraphedTrainStep

dev = torch.device("cuda:0")


# ------------------------------------------------------------------------------------------ stock architectures
class _StockDenseLayer(tnn.Module):
    def __init__(self, cin, growth=32, bn_size=4):
        super().__init__()
        self.f = tnn.Sequential(tnn.BatchNorm2d(cin), tnn.ReLU(), tnn.Conv2d(cin, bn_size * growth, 1, bias=False),
                                tnn.BatchNorm2d(bn_size * growth), tnn.ReLU(),
                                tnn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False))

    def forward(self, feats):
        return self.f(torch.cat(feats, 1))


class StockCNN(tnn.Module):
    def __init__(self, growth=32, blocks=2, layers=2, classes=6):
        super().__init__()
        c = 2 * growth
        self.stem = tnn.Sequential(tnn.Conv2d(3, c, 7, 2, 3, bias=False), tnn.BatchNorm2d(c), tnn.ReLU(),
                                   tnn.MaxPool2d(3, 2, 1))
        self.blocks = tnn.ModuleList()
        self.trans = tnn.ModuleList()
        for b in range(blocks):
            self.blocks.append(tnn.ModuleList(_StockDenseLayer(c + i * growth, growth) for i in range(layers)))
            c += layers * growth
            if b != blocks - 1:
                self.trans.append(tnn.Sequential(tnn.BatchNorm2d(c), tnn.ReLU(), tnn.Conv2d(c, c // 2, 1, bias=False),
                                                 tnn.AvgPool2d(2, 2)))
                c //= 2
        self.fc = tnn.Linear(c, classes)

    def forward(self, x):
        x = self.stem(x)
        for b, block in enumerate(self.blocks):
            feats = [x]
            for layer in block:
                feats.append(layer(feats))
            x = torch.cat(feats, 1)
            if b < len(self.trans):
                x = self.trans[b](x)
        return self.fc(x.mean((2, 3)))


class StockLSTM(tnn.Module):
    def __init__(self, hidden=128, classes=5, history=10, features=32):
        super().__init__()
        self.conv = tnn.Conv1d(history, 64, 1)
        self.lstm = tnn.LSTM(features, hidden, batch_first=True)
        self.fc = tnn.Linear(hidden, classes)

    def forward(self, x):
        x = torch.relu(self.conv(x))
        _, (h, _) = self.lstm(x)
        return self.fc(h.squeeze(0))


def stock_mlp():
    return tnn.Sequential(tnn.Linear(48, 38), tnn.ReLU(), tnn.Linear(38, 38), tnn.ReLU(), tnn.Linear(38, 5))


# ------------------------------------------------------------------------------------------ harness
def timeit(fn, steps):
    import gc
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    gc.collect()
    gc.freeze()             # same for ours and stock: no full-GC pause (70-115 ms) inside a timed loop of 1-10 ms steps
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    gc.unfreeze()
    return dt


def workload(name, B):
    g = torch.Generator().manual_seed(0)
    if name == "cnn":
        x = torch.randn(B, 3, 64, 64, generator=g).to(dev)
        y = torch.randint(0, 6, (B,), generator=g).to(dev)
        return x, y, SoftmaxCrossEntropyLoss, tnn.CrossEntropyLoss, "sgd", StockCNN
    if name == "lstm":
        x = torch.randn(B, 10, 32, generator=g).to(dev)
        y = torch.randn(B, 5, generator=g).to(dev)
        from distributed_deep_learning_amd.nn import L1Loss
        return x, y, L1Loss, tnn.L1Loss, "adam", StockLSTM
    x = torch.randn(B, 48, generator=g).to(dev)
    y = torch.randint(0, 5, (B,), generator=g).to(dev)
    return x, y, SoftmaxCrossEntropyLoss, tnn.CrossEntropyLoss, "sgd", stock_mlp


def ours(name, x, y, crit_cls, opt_kind, steps, graph):
    torch.manual_seed(0)
    m = models.build(name).to(dev)
    arena = ParamArena(m.named_parameters())
    opt = FusedAdam(arena, lr=1e-3) if opt_kind == "adam" else FusedSGD(arena, lr=0.01, momentum=0.9)
    crit = crit_cls()
    if graph:
        gs = GraphedTrainStep(m, crit, opt, x, y)
        return timeit(lambda: gs(x, y), steps)

    def step():
        loss = crit(m(x), y)
        loss.backward()
        opt.step()
    return timeit(step, steps)


def stock(x, y, crit_cls, opt_kind, factory, steps, amp=False):
    torch.manual_seed(0)
    m = factory().to(dev)
    if amp and x.dim() == 4:
        m = m.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3) if opt_kind == "adam" else \
        torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
    crit = crit_cls()

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            loss = crit(m(x).float(), y)
        loss.backward()
        opt.step()
    return timeit(step, steps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="32,512,4096")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--workloads", default="cnn,lstm,mlp")
    ap.add_argument("--ours-only", action="store_true", help="skip the stock columns (profiling runs)")
    a = ap.parse_args()
    print(f"{'workload':>8} {'batch':>6} | {'ours eager':>18} | {'ours hipGraph':>18} | {'stock fp32 eager':>18} | "
          f"{'stock bf16 autocast':>19} | vs fp32 | vs bf16", flush=True)
    for name in a.workloads.split(","):
        for B in map(int, a.batches.split(",")):
            x, y, crit_o, crit_s, opt_kind, factory = workload(name, B)
            te = ours(name, x, y, crit_o, opt_kind, a.steps, False)
            try:
                tg = ours(name, x, y, crit_o, opt_kind, a.steps, True)
            except Exception as e:                     # noqa: BLE001 - report and go on
                print(f"   graph capture failed for {name}: {type(e).__name__}: {e}", flush=True)
                tg = float("nan")
            if a.ours_only:
                print(f"{name:>8} {B:>6} | eager {te:7.3f} ms | hipGraph {tg:7.3f} ms", flush=True)
                continue
            ts = stock(x, y, crit_s, opt_kind, factory, a.steps)
            try:
                tb = stock(x, y, crit_s, opt_kind, factory, a.steps, amp=True)
            except Exception as e:                     # noqa: BLE001 - e.g. MIOpen without a bf16 RNN path
                print(f"   stock bf16 autocast failed for {name}: {type(e).__name__}: {e}", flush=True)
                tb = float("nan")
            best = min(t for t in (te, tg) if t == t)

            def f(t):
                return f"{t:7.3f} ms {B / t * 1e3:8.0f}/s"
            print(f"{name:>8} {B:>6} | {f(te)} | {f(tg)} | {f(ts)} | {f(tb)}  | {ts / best:5.2f}x  | {tb / best:5.2f}x", flush=True)


if __name__ == "__main__":
    main()
