"""Layers and fused functional units (GPU path = hand-written gfx950 kernels)."""
from .modules import (AvgPool2d, BatchNorm2d, Conv2d, Flatten, GlobalAvgPool2d, L1Loss, Linear, MaxPool2d, ReLU,
                      SoftmaxCrossEntropyLoss, conv_bn_act, residual_block, to_device_input)
from .rnn import LSTM

__all__ = ["AvgPool2d", "BatchNorm2d", "Conv2d", "Flatten", "GlobalAvgPool2d", "L1Loss", "LSTM", "Linear", "MaxPool2d",
           "ReLU", "SoftmaxCrossEntropyLoss", "conv_bn_act", "residual_block", "to_device_input"]
