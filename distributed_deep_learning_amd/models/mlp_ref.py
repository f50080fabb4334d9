"""The reference's MLP workload: Linear(in->38)+ReLU, ``hidden_layers`` x [Linear(38->38)+ReLU], Linear(38->5)
(``/root/reference/src/pytorch/MLP/model.py:51-59``); 48 input features for the MQTT data
(``/root/reference/src/pytorch/MLP/dataset.py:33-37``).  Emits logits (the reference's trailing Softmax in front
of CrossEntropyLoss is defect A5); 3,539 parameters at the default depth.  On the GPU the whole stack runs as ONE
forward and ONE backward launch (``nn.functional.FusedMLP`` -> ``csrc/mlp.hip``); the staged (model-parallel /
pipeline) paths and CPU tensors go layer by layer."""
from __future__ import annotations

import torch
import torch.nn as tnn

from .. import nn as dnn
from ..nn import functional as _F
from ..ops import kernels as _K


class MLPRef(tnn.Module):
    def __init__(self, input_size=48, hidden_layers=1, hidden_size=38, classes=5):
        super().__init__()
        if hidden_layers < 1:
            raise ValueError("Model requires at least one hidden layer")
        dims = [input_size] + [hidden_size] * (hidden_layers + 1)
        self.hidden = tnn.ModuleList(dnn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))
        self.out = dnn.Linear(hidden_size, classes)
        # the reference ends in Softmax, or in Sigmoid when classes < 2 (MLP/model.py:59).  The Softmax in front of
        # CrossEntropyLoss is defect A5 (this model emits logits); the single-output Sigmoid head is kept.
        self.sigmoid_head = classes < 2

    def _head(self, x):
        y = self.out(x)
        return torch.sigmoid(y.float()) if self.sigmoid_head else y

    def stages(self):
        return [lambda x, l=l: l(x, relu=True) for l in self.hidden] + [self._head]

    def stage_modules(self):
        from ..parallel.mp import FnStage
        return [FnStage(f, m) for f, m in zip(self.stages(), [*self.hidden, self.out])]

    def forward(self, x):
        layers = [*self.hidden, self.out]
        dims = [layers[0].in_features] + [l.out_features for l in layers]
        if x.is_cuda and x.dim() == 2 and _K.mlp_supported(dims) and all(l.weight.dtype == x.dtype == torch.float32
                                                                         for l in layers):
            # GPU: the whole stack is one forward and one backward launch (csrc/mlp.hip), fp32 like the reference
            params = []
            for l in layers:
                params += [l.weight, l.bias]
            y = _F.FusedMLP.apply(x, False, *params)
            return torch.sigmoid(y) if self.sigmoid_head else y
        for s in self.stages():
            x = s(x)
        return x


def mlp_ref(**kw):
    return MLPRef(**kw)
