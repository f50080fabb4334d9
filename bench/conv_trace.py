"""Phase breakdown of the ring conv kernel: per-block wall-clock stamps (prologue / K-loop / epilogue).

    python bench/conv_trace.py C H K R stride pad [f|d]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from distributed_deep_learning_amd.ops import _lib  # noqa: E402
from distributed_deep_learning_amd.ops import kernels as K  # noqa: E402

C, H, Kc, R, st, pad = [int(v) for v in sys.argv[1:7]]
mode = sys.argv[7] if len(sys.argv) > 7 else "f"
B = 256
dev = torch.device("cuda:0")
x = torch.randn(B, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(Kc, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
wk, wt = K.filter_repack(w)
y, _ = K.conv2d_fwd(x, wk, st, pad, want_stats=True)
dy = torch.randn_like(y)
lib = _lib.require()
lib.ddl_conv_set_trace.argtypes = [ctypes.c_void_p]
lib.ddl_conv_set_trace.restype = None
buf = torch.zeros(1 << 16, 4, dtype=torch.int64, device=dev)


def run():
    if mode == "f":
        K.conv2d_fwd(x, wk, st, pad, want_stats=True)
    else:
        K.conv2d_dgrad(dy, wt, (H, H), st, pad)


for _ in range(3):
    run()
torch.cuda.synchronize()
lib.ddl_conv_set_trace(ctypes.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
lib.ddl_conv_set_trace(None)
t = buf.cpu()
t = t[t[:, 3] > 0].double()
tick = 10.0                                        # wall_clock64: 100 MHz -> 10 ns
t0 = t[:, 0].min()
print(f"blocks {len(t)}  kernel span {(t[:, 3].max() - t0) * tick / 1e3:.1f} us")
for name, a, b in (("prologue (decode + first 2 stages issued)", 0, 1), ("K loop", 1, 2), ("epilogue (+store drain)", 2, 3),
                   ("block total", 0, 3)):
    d = (t[:, b] - t[:, a]) * tick / 1e3
    print(f"  {name:44s} mean {d.mean():7.2f} us  p10 {d.quantile(0.1):7.2f}  p90 {d.quantile(0.9):7.2f}")
start = (t[:, 0] - t0) * tick / 1e3
print(f"  block start times: p50 {start.quantile(0.5):.1f} us  p90 {start.quantile(0.9):.1f} us  max {start.max():.1f} us")
