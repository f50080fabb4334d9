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
        # conv+ReLU, MaxPool1d(1)+ReLU == ReLU.  A kernel-size-1 Conv1d is a Linear over the channel axis: computed as
        # ONE GEMM [B*32, 10] x [10, 64] (MIOpen's Conv1d costs seconds of kernel search / compilation on first use,
        # which dominated short runs of this workload; a broadcast matmul turns into B tiny GEMMs); the Conv1d module
        # stays the parameter container (state_dict keys).
        x = x.float()
        if x.is_cuda and x.shape[0] < 1024:       # big batches amortise MIOpen's start-up and its conv is 0.3 ms faster
            y = torch.nn.functional.linear(x.transpose(1, 2), self.conv.weight[:, :, 0], self.conv.bias)   # [B, 32, 64]
            return torch.relu(y).transpose(1, 2)                                                           # [B, 64, 32]
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
