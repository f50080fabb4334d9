"""cProfile of the host side of a training step (where does the enqueue time go?).

    python bench/host_profile.py [batch] [resnet50|cnn|mlp|lenet5]
"""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, ".")
from distributed_deep_learning_amd import models  # noqa: E402
from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss  # noqa: E402
from distributed_deep_learning_amd.optim import FusedSGD  # noqa: E402
from distributed_deep_learning_amd.parallel import ParamArena  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
name = sys.argv[2] if len(sys.argv) > 2 else "resnet50"
model = models.build(name).to(dev)
arena = ParamArena(model.named_parameters())
opt = FusedSGD(arena, lr=0.01, momentum=0.9)
crit = SoftmaxCrossEntropyLoss()
shapes = {"resnet50": ((3, 224, 224), 1000), "cnn": ((3, 64, 64), 6), "mlp": ((48,), 5), "lenet5": ((1, 28, 28), 10),
          "resnet18_cifar": ((3, 32, 32), 10)}
shape, classes = shapes[name]
x = torch.randn(B, *shape, device=dev)
y = torch.randint(0, classes, (B,), device=dev)


def step():
    loss = crit(model(x), y)
    loss.backward()
    opt.step()


import time  # noqa: E402

for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"20 steps: host enqueue {(t1 - t0) / 20 * 1e3:.3f} ms/step, wall {(t2 - t0) / 20 * 1e3:.3f} ms/step", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
