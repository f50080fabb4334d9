"""Process-wide counters shared by the optimizers and the layer weight caches.

The fused optimizers update master weights through raw device pointers, which torch's per-tensor version
counters cannot see; they call :func:`bump` after every step so cached bf16 filter repacks are refreshed.
"""
_weights_epoch = 0


def bump() -> int:
    global _weights_epoch
    _weights_epoch += 1
    return _weights_epoch


def weights_epoch() -> int:
    return _weights_epoch
