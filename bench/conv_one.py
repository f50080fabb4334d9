"""Run one conv shape a few times (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_deep_learning_amd.ops import kernels as K
C, H, Kc, R, st, pad = [int(v) for v in sys.argv[1:7]]
mode = sys.argv[7] if len(sys.argv) > 7 else "f"
B = int(os.environ.get('DDL_ONE_B', '256'))
dev = torch.device("cuda:0")
x = torch.randn(B, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(Kc, C, R, R, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
wk, wt = K.filter_repack(w)
y, _ = K.conv2d_fwd(x, wk, st, pad)
dy = torch.randn_like(y)
for _ in range(5):
    if mode == "f": K.conv2d_fwd(x, wk, st, pad, want_stats=True)
    elif mode == "d": K.conv2d_dgrad(dy, wt, (H, H), st, pad)
    else: K.conv2d_wgrad(dy, x, R, R, st, pad)
torch.cuda.synchronize()
