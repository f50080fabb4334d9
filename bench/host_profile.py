"""cProfile of the host side of the ResNet-50 training step (where does the enqueue time go?)."""
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
model = models.build("resnet50").to(dev)
arena = ParamArena(model.named_parameters())
opt = FusedSGD(arena, lr=0.01, momentum=0.9)
crit = SoftmaxCrossEntropyLoss()
x = torch.randn(B, 3, 224, 224, device=dev)
y = torch.randint(0, 1000, (B,), device=dev)


def step():
    loss = crit(model(x), y)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
