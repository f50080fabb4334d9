"""Peak device memory of the headline step at a given per-GPU batch: python bench/mem_probe.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_deep_learning_amd import models
from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss
from distributed_deep_learning_amd.optim import FusedSGD
from distributed_deep_learning_amd.parallel import ParamArena

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
m = models.build("resnet50").to(dev)
a = ParamArena(m.named_parameters())
o = FusedSGD(a, lr=0.01, momentum=0.9)
c = SoftmaxCrossEntropyLoss()
x = torch.randn(B, 3, 224, 224, device=dev)
y = torch.randint(0, 1000, (B,), device=dev)
for _ in range(3):
    c(m(x), y).backward()
    o.step()
torch.cuda.synchronize()
print(f"B={B}: peak allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, reserved "
      f"{torch.cuda.max_memory_reserved() / 2**30:.1f} GiB of {torch.cuda.get_device_properties(0).total_memory / 2**30:.0f} GiB")
