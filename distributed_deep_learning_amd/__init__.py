"""distributed_deep_learning_amd — an MI355X-native (gfx950 / CDNA4) data-parallel training framework.

Hot path: hand-written HIP kernels (``csrc/``: MFMA implicit-GEMM conv fwd/dgrad/wgrad, fused BN+ReLU,
softmax-cross-entropy, flat multi-tensor SGD/LARS/Adam) loaded in-tree via ctypes; gradients live in one flat
arena and are reduced in place with RCCL (all-reduce, or reduce-scatter/all-gather in the PS-style mode)
overlapped with backward; launcher / trainer / sampler / checkpoint cover the reference's capabilities
(``Belegkarnil/distributed-deep-learning``) with its defects fixed.  See README.md and SURVEY.md.
"""
__version__ = "0.1.0"

from . import config, data, dist, models, nn, ops, optim, parallel  # noqa: F401
from .config import Config  # noqa: F401
