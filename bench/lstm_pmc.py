"""A few fused-LSTM forward + backward launches at a fixed shape, for counter collection (bench/pmc.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_deep_learning_amd import nn as dnn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = dnn.LSTM(32, 128).cuda()
x = torch.randn(B, 64, 32, device="cuda", requires_grad=True)
for _ in range(4):
    out, _ = m(x)
    out.float().sum().backward()
torch.cuda.synchronize()
print("done")
