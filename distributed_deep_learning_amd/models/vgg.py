"""VGG-16 (configuration D) with torchvision-compatible keys (``features.N.weight``, ``classifier.N.weight``).

North-star config 4: 138 M parameters -> a 553 MB fp32 gradient, the bandwidth-bound xGMI all-reduce case.  Every
conv runs with bias + ReLU fused into the implicit-GEMM epilogue; the three classifier layers run on the framework's
own split-M MFMA GEMM with bias + ReLU + dropout fused around it (``nn.functional.LinearAct``).  ``batch_norm=True`` gives VGG-16-BN using the fused conv+BN+ReLU unit.
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from .. import nn as dnn
from ..nn.modules import conv_bn_act, conv_relu_chain, to_device_input

CFG_D = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
CFG_A = [64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"]


class VGG(tnn.Module):
    def __init__(self, cfg=CFG_D, num_classes: int = 1000, batch_norm: bool = False, input_size: int = 224,
                 hidden: int = 4096, dropout: float = 0.5):
        super().__init__()
        layers, c = [], 3
        for v in cfg:
            if v == "M":
                layers.append(dnn.MaxPool2d(2, 2, 0))
            else:
                layers.append(dnn.Conv2d(c, v, 3, 1, 1, bias=not batch_norm))
                if batch_norm:
                    layers.append(dnn.BatchNorm2d(v))
                layers.append(dnn.ReLU(inplace=True))
                c = v
        self.features = tnn.Sequential(*layers)
        self.batch_norm = batch_norm
        spatial = input_size // (2 ** sum(1 for v in cfg if v == "M"))
        self.flatten = dnn.Flatten()
        self.classifier = tnn.Sequential(
            dnn.Linear(c * spatial * spatial, hidden), dnn.ReLU(True), tnn.Dropout(dropout),
            dnn.Linear(hidden, hidden), dnn.ReLU(True), tnn.Dropout(dropout),
            dnn.Linear(hidden, num_classes))
        for m in self.modules():
            if isinstance(m, tnn.Conv2d):
                w = torch.empty(m.weight.shape)
                tnn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
                with torch.no_grad():
                    m.weight.copy_(w)
                    if m.bias is not None:
                        m.bias.zero_()
            elif isinstance(m, tnn.Linear):
                tnn.init.normal_(m.weight, 0, 0.01)
                tnn.init.constant_(m.bias, 0)

    def _features(self, x, lo=0, hi=None):
        mods = list(self.features)[lo:hi]
        if not self.batch_norm:
            # conv+bias+ReLU(+pool) runs are one fused node: ReLU gating and bias gradients ride in the next dgrad
            units, i = [], 0
            while i < len(mods):
                pool = mods[i + 2] if i + 2 < len(mods) and isinstance(mods[i + 2], tnn.MaxPool2d) else None
                units.append((mods[i], pool))
                i += 3 if pool is not None else 2
            return conv_relu_chain(units, x)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, dnn.Conv2d):
                if self.batch_norm:
                    x = conv_bn_act(m, mods[i + 1], x, relu=True)
                    i += 3
                else:
                    x = m(x, relu=True)
                    i += 2
            else:
                x = m(x)
                i += 1
        return x

    def _classifier(self, x):
        """``classifier`` with every Linear -> ReLU -> Dropout run fused into one unit (the Sequential stays the
        parameter container, so the torchvision keys ``classifier.0/3/6`` are unchanged)."""
        mods = list(self.classifier)
        if not x.is_cuda:
            return self.classifier(x)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, dnn.Linear):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], (dnn.ReLU, tnn.ReLU))
                j = i + 2 if relu else i + 1
                drop = mods[j].p if j < len(mods) and isinstance(mods[j], tnn.Dropout) else 0.0
                x = m(x, relu=relu, dropout=drop)
                i = j + (1 if drop or (j < len(mods) and isinstance(mods[j], tnn.Dropout)) else 0)
            else:
                x = m(x)
                i += 1
        return x

    def stages(self):
        """Five conv blocks (split after each max-pool) + classifier, for inter-layer partitioning."""
        cuts, mods = [0], list(self.features)
        for i, m in enumerate(mods):
            if isinstance(m, tnn.MaxPool2d):
                cuts.append(i + 1)
        st = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            st.append(lambda x, a=a, b=b: self._features(to_device_input(x) if a == 0 else x, a, b))
        st.append(lambda x: self._classifier(self.flatten(x)))
        return st

    def stage_modules(self):
        from ..parallel.mp import FnStage
        cuts, mods = [0], list(self.features)
        for i, m in enumerate(mods):
            if isinstance(m, tnn.MaxPool2d):
                cuts.append(i + 1)
        fns = self.stages()
        out = [FnStage(f, *mods[a:b]) for f, a, b in zip(fns[:-1], cuts[:-1], cuts[1:])]
        out.append(FnStage(fns[-1], self.classifier))
        return out

    def forward(self, x):
        x = to_device_input(x)
        x = self._features(x)
        return self._classifier(self.flatten(x))


def vgg16(num_classes=1000, **kw):
    return VGG(CFG_D, num_classes, **kw)


def vgg16_bn(num_classes=1000, **kw):
    return VGG(CFG_D, num_classes, batch_norm=True, **kw)


def vgg11(num_classes=1000, **kw):
    return VGG(CFG_A, num_classes, **kw)
