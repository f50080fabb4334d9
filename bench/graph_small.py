"""Eager vs HIP-graph step time for the launch-bound small models.

    python bench/graph_small.py [model] [batch]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from distributed_deep_learning_amd import models  # noqa: E402
from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss  # noqa: E402
from distributed_deep_learning_amd.optim import FusedSGD  # noqa: E402
from distributed_deep_learning_amd.parallel import ParamArena  # noqa: E402
from distributed_deep_learning_amd.utils import GraphedTrainStep  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "resnet18_cifar"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
shape = {"resnet18_cifar": (3, 32, 32), "lenet5": (1, 28, 28), "cnn": (3, 64, 64)}[name]
classes = {"resnet18_cifar": 10, "lenet5": 10, "cnn": 6}[name]
dev = torch.device("cuda:0")
x = torch.randn(B, *shape, device=dev)
y = torch.randint(0, classes, (B,), device=dev)


def make():
    torch.manual_seed(0)
    m = models.build(name).to(dev)
    a = ParamArena(m.named_parameters())
    return m, FusedSGD(a, lr=0.01, momentum=0.9), SoftmaxCrossEntropyLoss()


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


m, o, c = make()


def eager():
    loss = c(m(x), y)
    loss.backward()
    o.step()


te = timeit(eager)
m2, o2, c2 = make()
g = GraphedTrainStep(m2, c2, o2, x, y)
tg = timeit(lambda: g(x, y))
print(f"{name} B={B}: eager {te:.3f} ms/step ({B / te * 1e3:.0f} img/s) | hipGraph {tg:.3f} ms/step "
      f"({B / tg * 1e3:.0f} img/s) | speedup {te / tg:.2f}x", flush=True)
