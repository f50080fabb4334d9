import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from test_gpu_round2 import _dense_layers, _dense_block_fp32, _rand_act, _dev
from distributed_deep_learning_amd.nn import modules as M

dev = _dev()
N, c0, HW, growth, nl = 4, 64, 16, 32, 2
layers = _dense_layers(c0, growth, nl, 4, dev, 5)
params = [p for l in layers for m in l for p in m.parameters()]
names = [f"L{i}.{n}.{pn}" for i, l in enumerate(layers) for n, m in zip(("n1", "c1", "n2", "c2"), l) for pn, _ in m.named_parameters()]
x0 = _rand_act((N, c0, HW, HW), 21)
gy = _rand_act((N, c0 + growth * nl, HW, HW), 22, 0.5)

def run(fn, x, g):
    for p in params: p.grad = None
    x = x.clone().requires_grad_(True)
    y = fn(x)
    y.backward(g)
    return y.detach().float(), x.grad.float(), [p.grad.clone().float() for p in params]

fused = run(lambda x: M.dense_block(layers, x), x0, gy)
def layerwise(x):
    feats = [x]
    for n1, c1, n2, c2 in layers:
        h = torch.cat(feats, 1) if len(feats) > 1 else feats[0]
        h = n1(h, relu=True)
        h = M.conv_bn_act(c1, n2, h, relu=True)
        feats.append(c2(h))
    return torch.cat(feats, 1)
lw = run(layerwise, x0, gy)
ref = run(lambda x: _dense_block_fp32(layers, x), x0.float(), gy.float())
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
print("y      fused-ref %.4f  lw-ref %.4f  fused-lw %.4f" % (rel(fused[0], ref[0]), rel(lw[0], ref[0]), rel(fused[0], lw[0])))
print("dx     fused-ref %.4f  lw-ref %.4f  fused-lw %.4f" % (rel(fused[1], ref[1]), rel(lw[1], ref[1]), rel(fused[1], lw[1])))
for n, a, b, c in zip(names, fused[2], lw[2], ref[2]):
    print("%-14s fused-ref %.4f  lw-ref %.4f  fused-lw %.4f" % (n, rel(a, c), rel(b, c), rel(a, b)))
