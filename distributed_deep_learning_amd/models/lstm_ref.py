"""The reference's LSTM workload (``/root/reference/src/pytorch/LSTM/model.py:70-94``): Conv1d(10->64, k=1)+ReLU,
MaxPool1d(1)+ReLU (an identity), ``hidden_layers`` stacked LSTMs over the 64 conv channels as the time axis
(input size 32 -> hidden), final hidden state -> Linear(hidden->5).  Trained with L1 loss + Adam
(``/root/reference/src/pytorch/LSTM/main.py:163-164``).  On the GPU every LSTM layer is ONE launch of the fused
recurrent kernel (``nn.LSTM`` -> ``csrc/lstm.hip``: weights in registers, all 64 steps inside the kernel); CPU tensors
and unsupported widths take ``torch.nn.LSTM``.  84,293 parameters at the defaults."""
from __future__ import annotations

import torch
import torch.nn as tnn

from .. import nn as dnn
from ..nn import functional as _F
from ..ops import kernels as _K


class LSTMRef(tnn.Module):
    def __init__(self, hidden_layers=1, hidden_params=128, classes=5, history=10, features=32):
        super().__init__()
        if hidden_layers < 1:
            raise ValueError("Model requires at least one hidden layer")
        self.conv = tnn.Conv1d(history, 64, 1)
        self.lstms = tnn.ModuleList(
            dnn.LSTM(features if i == 0 else hidden_params, hidden_params, batch_first=True) for i in range(hidden_layers))
        self.fc = dnn.Linear(hidden_params, classes)

    def _front(self, x):
        # conv+ReLU, MaxPool1d(1)+ReLU == ReLU.  GPU: one VALU kernel (the reduction is only `history` = 10 deep) that
        # writes bf16 in the [B, T=64, I=32] layout the fused LSTM kernel reads, weight / bias gradients straight into the
        # arena (nn.functional.Conv1dK1Relu); the Conv1d module stays the parameter container (state_dict keys).
        x = x.float()
        if (x.is_cuda and not x.requires_grad and x.is_contiguous()
                and _K.conv1d_k1_supported(x.shape[1], self.conv.out_channels, x.shape[2])):
            return _F.Conv1dK1Relu.apply(x, self.conv.weight, self.conv.bias)
        return torch.relu(self.conv(x))

    def _lstm(self, i, x):
        out, (h, _c) = self.lstms[i](x)
        return out if i + 1 < len(self.lstms) else h.squeeze(0).float()

    def stages(self):
        return [self._front] + [lambda x, i=i: self._lstm(i, x) for i in range(len(self.lstms))] + [self.fc]

    def stage_modules(self):
        from ..parallel.mp import FnStage
        return [FnStage(f, m) for f, m in zip(self.stages(), [self.conv, *self.lstms, self.fc])]

    def forward(self, x):
        for s in self.stages():
            x = s(x)
        return x


def lstm_ref(**kw):
    return LSTMRef(**kw)
