"""ctypes binding to ``lib/libddl_hip.so`` (the hand-written gfx950 kernel library).

The library has a C ABI: raw device pointers, plain ints/floats and a ``hipStream_t``.  Every wrapper launches
on torch's *current* HIP stream, so the kernels order correctly with ATen ops, with the RCCL communication
streams of ``torch.distributed`` and inside ``torch.cuda.graph`` capture.

There is deliberately no silent fallback: on a GPU box a missing library is an error (``require()``).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch

_PKG = Path(__file__).resolve().parent.parent
# DDL_LIB: another build of the library (a variant with parts compiled out, bench/README.md) instead of the in-tree one
LIB_PATH = Path(os.environ["DDL_LIB"]) if os.environ.get("DDL_LIB") else _PKG / "lib" / "libddl_hip.so"

_lib = None
_load_error: Exception | None = None

_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong

_SIGS = {
    "ddl_conv_igemm": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp] + [_i] * 13 + [_vp],
    "ddl_conv_wgrad": [_vp, _vp, _vp] + [_i] * 11 + [_vp, _ll, _vp],
    "ddl_conv_wgrad_plan": [_i] * 11,
    "ddl_wgrad_halo_min_stages": [_i],
    "ddl_wgrad_halo_launches": [],
    "ddl_conv_c64_min_stages": [_i],
    "ddl_conv_c64_launches": [],
    "ddl_relu_bwd_bias_grad": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ddl_spin": [_ll, _vp],
    "ddl_stream_proxy": [_vp, _ll, C.c_int, C.c_int, _vp],
    "ddl_cast_ranges": [_vp, _vp, _vp, _i, _ll, _vp],
    "ddl_pack_owned": [_vp, _vp, _vp, _vp, _ll, _vp],
    "ddl_scatter_values": [_vp, _vp, _vp, _ll, _vp],
    "ddl_bn_slots": [],
    "ddl_slab_fold": [_vp, _i, _vp, _i, _vp],
    "ddl_bn_stats": [_vp, _vp, _i, _i, _vp],
    "ddl_bn_stats_ld": [_vp, _i, _vp, _i, _i, _vp],
    "ddl_bn_apply_ld": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ddl_bn_backward_ld": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ddl_slab_fold_totals": [_vp, _i, _vp, _i, _i, _vp],
    "ddl_bn_finalize_totals": [_vp, _i, _i, _f, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "ddl_copy_channels": [_vp, _i, _vp, _i, _ll, _i, _vp],
    "ddl_conv_fwd_bn": [_vp, _vp, _vp, _i, _vp] + [_i] * 11 + [_vp, _vp, _vp, _vp, _f, _f, _vp, _vp],
    "ddl_conv_fwd_ld": [_vp, _vp, _vp, _i, _vp] + [_i] * 11 + [_vp],
    "ddl_bn_finalize": [_vp, _i, _i, _f, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "ddl_bn_apply": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ddl_bn_backward": [_vp] * 12 + [_i, _i, _i, _i, _i, _vp],
    "ddl_bn_backward_pair": [_vp] * 20 + [_i, _i, _i, _i, _vp],
    "ddl_bn_relu_maxpool_fwd": [_vp] * 7 + [_i] * 9 + [_vp],
    "ddl_bn_relu_maxpool_bwd": [_vp] * 13 + [_i] * 10 + [_vp],
    "ddl_maxpool_fwd": [_vp, _vp, _vp] + [_i] * 9 + [_vp],
    "ddl_maxpool_bwd": [_vp, _vp, _vp] + [_i] * 9 + [_vp],
    "ddl_avgpool_fwd": [_vp, _vp] + [_i] * 8 + [_vp],
    "ddl_avgpool_bwd": [_vp, _vp] + [_i] * 8 + [_vp],
    "ddl_gap_fwd": [_vp, _vp, _i, _i, _i, _vp],
    "ddl_gap_bwd": [_vp, _vp, _i, _i, _i, _vp],
    "ddl_cast_f32_bf16": [_vp, _vp, _ll, _vp],
    "ddl_cast_bf16_f32": [_vp, _vp, _ll, _vp],
    "ddl_relu_fwd": [_vp, _vp, _ll, _vp],
    "ddl_relu_bwd": [_vp, _vp, _vp, _ll, _vp],
    "ddl_add": [_vp, _vp, _vp, _ll, _i, _vp],
    "ddl_bias_grad": [_vp, _vp, _i, _i, _vp],
    "ddl_filter_repack": [_vp, _i, _vp, _vp, _i, _i, _i, _vp],
    "ddl_filter_transpose_batched": [_vp, _vp, _vp, _i, _i, _vp],
    "ddl_repack_desc_bytes": [],
    "ddl_nchw_to_nhwc_pad": [_vp, _vp, _i, _i, _i, _i, _vp],
    "ddl_nchw_to_s2d": [_vp, _vp, _i, _i, _i, _vp],
    "ddl_softmax_xent": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "ddl_chunk_elems": [],
    "ddl_flat_sumsq": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ddl_flat_sgd": [_vp] * 9 + [_i, _i, _i, _vp],
    "ddl_flat_adam": [_vp] * 8 + [_i, _i, _i, _vp],
    "ddl_linear_epilogue": [_vp, _vp, _vp, _i, _i, _i, _f, _vp, _i, _vp],
    "ddl_linear_bwd_gate": [_vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _vp],
    "ddl_transpose_pad": [_vp, _i, _vp, _i, _i, _i, _ll, _vp],
    "ddl_cast_rows": [_vp, _vp, _i, _i, _ll, _i, _vp],
    "ddl_l1_loss": [_vp, _i, _vp, _vp, _vp, _ll, _vp],
    "ddl_conv1d_k1_relu_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ddl_conv1d_k1_relu_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ddl_comm_load": [C.c_char_p],
    "ddl_comm_version": [],
    "ddl_comm_unique_id": [_vp],
    "ddl_comm_init": [_vp, _i, _i, _vp],
    "ddl_comm_destroy": [_vp],
    "ddl_comm_abort": [_vp],
    "ddl_comm_count": [_vp],
    "ddl_comm_user_rank": [_vp],
    "ddl_comm_all_reduce": [_vp, _vp, _vp, _ll, _i, _i, _vp],
    "ddl_comm_reduce_scatter": [_vp, _vp, _vp, _ll, _i, _i, _vp],
    "ddl_comm_all_gather": [_vp, _vp, _vp, _ll, _i, _vp],
    "ddl_comm_broadcast": [_vp, _vp, _ll, _i, _i, _vp],
    "ddl_mlp_fwd": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp],
    "ddl_mlp_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp],
    "ddl_lstm_supported": [_i, _i],
    "ddl_lstm_fwd": [_vp, _ll, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ddl_lstm_bwd": [_vp, _ll, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ddl_rows_gemm_nt": [_vp, _vp, _vp, _i, _i, _i, _vp],
}


_LONG_RESULT = {"ddl_conv_wgrad_plan", "ddl_conv_c64_launches", "ddl_wgrad_halo_launches"}


def _load():
    global _lib, _load_error
    if _lib is not None or _load_error is not None:
        return _lib
    try:
        lib = C.CDLL(str(LIB_PATH))
        for name, argtypes in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = _ll if name in _LONG_RESULT else _i
        _lib = lib
    except Exception as e:  # noqa: BLE001 - recorded and re-raised by require()
        _load_error = e
    return _lib


def available() -> bool:
    """True when the kernel library is loadable (it links libamdhip64, so this also works on a CPU box)."""
    return _load() is not None


def require():
    lib = _load()
    if lib is None:
        raise RuntimeError(
            f"HIP kernel library not loadable from {LIB_PATH}: {_load_error!r}. "
            "Build it with `python -m distributed_deep_learning_amd.build`."
        )
    return lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)       # noqa: SLF001
_cur_device = getattr(torch._C, "_cuda_getDevice", None)                 # noqa: SLF001


def stream_ptr() -> int:
    """hipStream_t of torch's current stream.  ``torch.cuda.current_stream()`` builds a Python Stream object and
    costs ~9 us per call (1.5 ms of host time per ResNet-50 step); the raw accessor is ~20x cheaper."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


class WrongDeviceError(RuntimeError):
    pass


def check_device(t: torch.Tensor, what: str = "kernel") -> None:
    """Every wrapper launches on the CURRENT device's current stream (``stream_ptr``): an operand that lives on
    another GPU would be processed by the wrong device, unordered against the ATen work on its own stream.  Callers
    that span devices (model / pipeline parallelism, per-stage optimizers) wrap their work in :func:`device_of`."""
    if t.is_cuda and _cur_device is not None and t.device.index != _cur_device():
        raise WrongDeviceError(f"{what}: operand on cuda:{t.device.index} but the current device is cuda:{_cur_device()} "
                               "(wrap the call in ops._lib.device_of(tensor))")


class device_of:
    """``with device_of(t): ...`` — make ``t``'s GPU the current device for the block (a no-op for CPU tensors and
    when it already is current, so the single-GPU hot path pays one integer compare)."""

    __slots__ = ("idx", "prev")

    def __init__(self, t):
        dev = t if isinstance(t, torch.device) else getattr(t, "device", None)
        self.idx = dev.index if (dev is not None and dev.type == "cuda") else None
        self.prev = None

    def __enter__(self):
        if self.idx is not None and _cur_device is not None:
            cur = _cur_device()
            if cur != self.idx:
                self.prev = cur
                torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


def ptr(t):
    """Device address of a tensor; ``None`` and raw integer addresses pass through (hot paths compute sub-buffer
    addresses by arithmetic instead of creating tensor views)."""
    if t is None or isinstance(t, int):
        return t
    return t.data_ptr()


class HipKernelError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise HipKernelError(f"{what} failed with code {rc}" + (" (bad arguments)" if rc < 0 else " (hipError_t)"))


def chunk_elems() -> int:
    """Arena alignment quantum (elements); constant mirrored from csrc/optim.hip so CPU boxes need no .so."""
    return 1024


def bn_slots() -> int:
    return 64
