"""Quick 1-GPU ResNet-50 step timing + parity check vs stock torch (dev tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_deep_learning_amd.models.resnet import resnet50
from distributed_deep_learning_amd import nn as dnn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = resnet50().to(dev)
crit = dnn.SoftmaxCrossEntropyLoss()
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
x = torch.randn(B, 3, 224, 224, device=dev)
y = torch.randint(0, 1000, (B,), device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(model(x), y)
    loss.backward()
    opt.step()
    return loss
for i in range(3):
    l = step()
    print("warm", i, float(l), flush=True)
torch.cuda.synchronize()
t0 = time.time()
for i in range(steps):
    l = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f"B={B} {dt*1e3:.2f} ms/step  {B/dt:.1f} img/s  loss {float(l):.4f}", flush=True)
