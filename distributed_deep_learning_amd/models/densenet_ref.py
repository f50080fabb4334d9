"""The reference's CNN workload: a DenseNet-BC variant for 64x64 RGB crops, 6 classes
(``/root/reference/src/pytorch/CNN/model.py:116-193``).

Same architecture — stem conv7x7/2 + BN + ReLU + maxpool3/2; ``dense_blocks`` blocks of ``dense_layers``
bottleneck layers (BN-ReLU-conv1x1(bn_size*growth)-BN-ReLU-conv3x3(growth), growth 32); transition
(BN-ReLU-conv1x1(C/2)-avgpool2) between blocks; pool; linear — and the same parameter census (208,838 parameters
in 32 tensors at the defaults), with the reference's defects fixed:

* emits **logits** (A5: the reference applies Softmax and then CrossEntropyLoss),
* global average pool instead of ``AvgPool2d(7)`` on an 8x8 map (A10),
* BN momentum is exposed, PyTorch default 0.1 (A7: reference uses 0.99; pass ``bn_momentum=0.99`` for parity),
* pre-activation BN+ReLU run as one fused kernel reading the block's single feature buffer in place: the features
  are never concatenated on the GPU (``model.py:43-47`` re-copies them for every layer), conv2's epilogue writes its
  growth channels into the buffer and supplies their batch statistics to every later layer.
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from .. import nn as dnn
from ..nn.modules import conv_bn_act, conv_bn_relu_pool, dense_block


class DenseLayer(tnn.Module):
    def __init__(self, cin, growth, bn_size, eps, momentum):
        super().__init__()
        self.norm1 = dnn.BatchNorm2d(cin, eps=eps, momentum=momentum)
        self.conv1 = dnn.Conv2d(cin, bn_size * growth, 1, 1, 0, bias=False)
        self.norm2 = dnn.BatchNorm2d(bn_size * growth, eps=eps, momentum=momentum)
        self.conv2 = dnn.Conv2d(bn_size * growth, growth, 3, 1, 1, bias=False)

    def forward(self, x):
        """The layer alone: ``x`` = every earlier feature map; returns ``x`` extended by this layer's channels."""
        return dense_block([(self.norm1, self.conv1, self.norm2, self.conv2)], x)


class DenseBlock(tnn.Module):
    def __init__(self, num_layers, cin, bn_size, growth, eps, momentum):
        super().__init__()
        self.layers = tnn.ModuleList(DenseLayer(cin + i * growth, growth, bn_size, eps, momentum)
                                     for i in range(num_layers))

    def forward(self, x):
        # GPU: one feature buffer, every layer writes its growth channels in place (nn.functional.DenseBlock)
        return dense_block([(l.norm1, l.conv1, l.norm2, l.conv2) for l in self.layers], x)


class Transition(tnn.Module):
    def __init__(self, cin, cout, eps, momentum):
        super().__init__()
        self.norm = dnn.BatchNorm2d(cin, eps=eps, momentum=momentum)
        self.conv = dnn.Conv2d(cin, cout, 1, 1, 0, bias=False)
        self.pool = dnn.AvgPool2d(2, 2)

    def forward(self, x):
        return self.pool(self.conv(self.norm(x, relu=True)))


class DenseNetRef(tnn.Module):
    def __init__(self, growth_rate=32, dense_blocks=2, dense_layers=2, bn_size=4, classes=6, bn_eps=1e-3,
                 bn_momentum=0.1):
        super().__init__()
        if dense_blocks < 1:
            raise ValueError("Model requires at least one dense block")
        c = growth_rate * 2
        self.conv0 = dnn.Conv2d(3, c, 7, 2, 3, bias=False)
        self.norm0 = dnn.BatchNorm2d(c, eps=bn_eps, momentum=bn_momentum)
        self.pool0 = dnn.MaxPool2d(3, 2, 1)
        blocks = []
        for i in range(dense_blocks):
            blocks.append(DenseBlock(dense_layers, c, bn_size, growth_rate, bn_eps, bn_momentum))
            c += dense_layers * growth_rate
            if i != dense_blocks - 1:
                blocks.append(Transition(c, c // 2, bn_eps, bn_momentum))
                c //= 2
        self.blocks = tnn.ModuleList(blocks)
        self.pool = dnn.GlobalAvgPool2d()
        self.classifier = dnn.Linear(c, classes)
        for m in self.modules():
            if isinstance(m, tnn.Conv2d):
                w = torch.empty(m.weight.shape)
                tnn.init.kaiming_normal_(w)
                with torch.no_grad():
                    m.weight.copy_(w)
            elif isinstance(m, tnn.BatchNorm2d):
                tnn.init.constant_(m.weight, 1)
                tnn.init.constant_(m.bias, 0)
            elif isinstance(m, tnn.Linear):
                tnn.init.constant_(m.bias, 0)

    def _stem(self, x):
        return conv_bn_relu_pool(self.conv0, self.norm0, self.pool0, x)       # converts the input itself

    def stages(self):
        """Logical layers for the model-parallel partitioner (the reference's layer ids, ``model.py:155-184``)."""
        return [self._stem, *self.blocks, lambda x: self.classifier(self.pool(x))]

    def stage_modules(self):
        from ..parallel.mp import FnStage
        fns = self.stages()
        return [FnStage(fns[0], self.conv0, self.norm0, self.pool0), *self.blocks,
                FnStage(fns[-1], self.pool, self.classifier)]

    def forward(self, x):
        for s in self.stages():
            x = s(x)
        return x


def densenet_ref(**kw):
    return DenseNetRef(**kw)
