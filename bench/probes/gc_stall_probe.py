"""Is the occasional ~100 ms step of a launch-bound model a Python garbage-collection pause?  Times every step on the
host, logs every gc run (generation, duration), then repeats with gc.freeze() after warm-up."""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from distributed_deep_learning_amd import models  # noqa: E402
from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss  # noqa: E402
from distributed_deep_learning_amd.optim import FusedSGD  # noqa: E402
from distributed_deep_learning_amd.parallel import ParamArena  # noqa: E402

dev = torch.device("cuda:0")
model = models.build("resnet18_cifar").to(dev)
arena = ParamArena(model.named_parameters())
opt = FusedSGD(arena, lr=0.01, momentum=0.9)
crit = SoftmaxCrossEntropyLoss()
x = torch.randn(512, 3, 32, 32, device=dev)
y = torch.randint(0, 10, (512,), device=dev)
log, t_start = [], [0.0]


def cb(phase, info):
    if phase == "start":
        t_start[0] = time.perf_counter()
    else:
        log.append((info["generation"], (time.perf_counter() - t_start[0]) * 1e3, info["collected"]))


gc.callbacks.append(cb)


def step():
    loss = crit(model(x), y)
    loss.backward()
    opt.step()


def run(tag, n=150):
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    log.clear()
    ts = []
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    marks[0].record()
    for i in range(n):
        t0 = time.perf_counter()
        step()
        marks[i + 1].record()
        ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    gpu = [marks[i].elapsed_time(marks[i + 1]) for i in range(n)]
    print(f"{tag}: GPU step median {sorted(gpu)[n // 2]:.2f} ms, GPU steps over 20 ms: {[(i, round(t, 1)) for i, t in enumerate(gpu) if t > 20]}",
          flush=True)
    slow = [(i, round(t, 1)) for i, t in enumerate(ts) if t > 20]
    g2 = [(g, round(d, 1), c) for g, d, c in log if d > 5]
    print(f"{tag}: median host step {sorted(ts)[n // 2]:.2f} ms, steps over 20 ms: {slow}, gc runs over 5 ms (gen, ms, collected): {g2}, "
          f"gc runs total {len(log)}, tracked objects {len(gc.get_objects())}", flush=True)


run("default gc")
gc.collect()
gc.freeze()
run("after gc.freeze()")
