"""ResNet family (He et al.) with torchvision-compatible ``state_dict`` keys (``conv1.weight``,
``layer1.0.bn2.running_var``, ``layer2.0.downsample.0.weight``, ``fc.bias`` ...).

North-star workloads (BASELINE.json): ResNet-50 on 224x224 (headline) and ResNet-18 with the CIFAR stem
(3x3 conv, no max-pool) on 32x32.  Every conv->BN->ReLU triple runs as one fused unit
(:func:`..nn.modules.conv_bn_act`); the bottleneck tail is conv -> BN -> +identity -> ReLU in a single BN-apply
kernel.  Stride sits on the 3x3 conv of the bottleneck (torchvision "v1.5").

The reference repo has no ResNet; its CNN workload (``/root/reference/src/pytorch/CNN/model.py``) lives in
:mod:`.densenet_ref`.
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from .. import nn as dnn
from ..nn.modules import conv_bn_act, conv_bn_relu_pool, residual_block, to_device_input


def _conv3x3(cin, cout, stride=1):
    return dnn.Conv2d(cin, cout, 3, stride, 1, bias=False)


def _conv1x1(cin, cout, stride=1):
    return dnn.Conv2d(cin, cout, 1, stride, 0, bias=False)


class BasicBlock(tnn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = dnn.BatchNorm2d(planes)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = dnn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        ds = (self.downsample[0], self.downsample[1]) if self.downsample is not None else None
        return residual_block([(self.conv1, self.bn1), (self.conv2, self.bn2)], ds, x)


class Bottleneck(tnn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv1x1(inplanes, planes)
        self.bn1 = dnn.BatchNorm2d(planes)
        self.conv2 = _conv3x3(planes, planes, stride)
        self.bn2 = dnn.BatchNorm2d(planes)
        self.conv3 = _conv1x1(planes, planes * 4)
        self.bn3 = dnn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        ds = (self.downsample[0], self.downsample[1]) if self.downsample is not None else None
        return residual_block([(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)], ds, x)


class ResNet(tnn.Module):
    def __init__(self, block, layers, num_classes=1000, cifar_stem=False, zero_init_residual=False):
        super().__init__()
        self.inplanes = 64
        self.cifar_stem = cifar_stem
        if cifar_stem:
            self.conv1 = dnn.Conv2d(3, 64, 3, 1, 1, bias=False)
        else:
            self.conv1 = dnn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = dnn.BatchNorm2d(64)
        self.maxpool = tnn.Identity() if cifar_stem else dnn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0], 1)
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.avgpool = dnn.GlobalAvgPool2d()
        self.fc = dnn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, tnn.Conv2d):
                w = torch.empty(m.weight.shape)
                tnn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
                with torch.no_grad():
                    m.weight.copy_(w)
            elif isinstance(m, tnn.BatchNorm2d):
                tnn.init.constant_(m.weight, 1)
                tnn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    tnn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlock):
                    tnn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = tnn.Sequential(_conv1x1(self.inplanes, planes * block.expansion, stride),
                                        dnn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return tnn.Sequential(*layers)

    # The stages below are what the model-parallel / pipeline wrappers split on.
    def stem(self, x):
        if self.cifar_stem:
            return conv_bn_act(self.conv1, self.bn1, to_device_input(x), relu=True)
        return conv_bn_relu_pool(self.conv1, self.bn1, self.maxpool, x)       # converts the input itself

    def head(self, x):
        return self.fc(self.avgpool(x))

    def stages(self):
        """Ordered list of callables (stem, layer1..4, head) for inter-layer partitioning."""
        return [self.stem, self.layer1, self.layer2, self.layer3, self.layer4, self.head]

    def stage_modules(self):
        from ..parallel.mp import FnStage
        return [FnStage(self.stem, self.conv1, self.bn1, self.maxpool), self.layer1, self.layer2, self.layer3,
                self.layer4, FnStage(self.head, self.avgpool, self.fc)]

    def forward(self, x):
        x = self.stem(x)
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        x = self.layer4(x)
        return self.head(x)


def resnet18(num_classes=1000, **kw):
    return ResNet(BasicBlock, [2, 2, 2, 2], num_classes, **kw)


def resnet18_cifar(num_classes=10, **kw):
    return ResNet(BasicBlock, [2, 2, 2, 2], num_classes, cifar_stem=True, **kw)


def resnet34(num_classes=1000, **kw):
    return ResNet(BasicBlock, [3, 4, 6, 3], num_classes, **kw)


def resnet50(num_classes=1000, **kw):
    return ResNet(Bottleneck, [3, 4, 6, 3], num_classes, **kw)


def resnet101(num_classes=1000, **kw):
    return ResNet(Bottleneck, [3, 4, 23, 3], num_classes, **kw)
