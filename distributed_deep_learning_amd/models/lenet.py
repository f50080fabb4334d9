"""LeNet-5 on MNIST-shape input (north-star config 1: the CPU / gloo plumbing model).

Classic layout: conv5x5(1->6, pad 2) - pool - conv5x5(6->16) - pool - fc120 - fc84 - fc10 (61,706 parameters).
Channel counts 1/6 are below the GPU kernels' 8-channel granularity: on a GPU the input is padded to 8 channels
and conv1 pads its 6 output channels physically to 8 (``Conv2d(pad_out=True)``, the extra ones stay zero through
ReLU and pooling), so both convolutions run on the MFMA kernels; on CPU (the gloo plumbing config) nothing changes.
"""
from __future__ import annotations

import torch.nn as tnn

from .. import nn as dnn


class LeNet5(tnn.Module):
    def __init__(self, num_classes: int = 10, in_channels: int = 1):
        super().__init__()
        self.conv1 = dnn.Conv2d(in_channels, 6, 5, 1, 2, pad_out=True)
        self.pool1 = dnn.MaxPool2d(2, 2, 0)
        self.conv2 = dnn.Conv2d(6, 16, 5, 1, 0)
        self.pool2 = dnn.MaxPool2d(2, 2, 0)
        self.flatten = dnn.Flatten()
        self.fc1 = dnn.Linear(16 * 5 * 5, 120)
        self.fc2 = dnn.Linear(120, 84)
        self.fc3 = dnn.Linear(84, num_classes)

    def stages(self):
        return [lambda x: self.pool1(self.conv1(x, relu=True)), lambda x: self.pool2(self.conv2(x, relu=True)),
                lambda x: self.fc1(self.flatten(x), relu=True), lambda x: self.fc2(x, relu=True), self.fc3]

    def stage_modules(self):
        from ..parallel.mp import FnStage
        owned = [(self.conv1, self.pool1), (self.conv2, self.pool2), (self.fc1,), (self.fc2,), (self.fc3,)]
        return [FnStage(f, *o) for f, o in zip(self.stages(), owned)]

    def forward(self, x):
        for s in self.stages():
            x = s(x)
        return x


def lenet5(num_classes=10, **kw):
    return LeNet5(num_classes, **kw)
