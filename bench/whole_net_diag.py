"""Whole-network parity table: our bf16 HIP model on the GPU and stock PyTorch under bf16 autocast on the CPU, both
against the SAME fp32 CPU model with shared weights (one training-mode forward + backward).

    python bench/whole_net_diag.py resnet50 32 [residual-branch gain]   |   python bench/whole_net_diag.py vgg16 8
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

import test_gpu_model as T  # noqa: E402

name, batch = sys.argv[1], int(sys.argv[2])
gain = float(sys.argv[3]) if len(sys.argv) > 3 else None
r = T._whole_network_parity(name, batch, 224, calibrate=True, branch_gain=gain)
print(f"{name} batch {batch}{'' if gain is None else f' residual-branch gain {gain}'}: logits rel err ours {r['logit_err']:.4f} stock-bf16 {r['logit_err_stock']:.4f}; "
      f"loss ours {r['loss_err']:.5f} stock-bf16 {r['loss_err_stock']:.5f}; "
      f"whole-gradient cosine ours {r['cos_all']:.5f} stock-bf16 {r['cos_all_stock']:.5f}")
print(f"{'parameter':44s} {'numel':>9s} {'err ours':>9s} {'err stock':>9s} {'cos ours':>9s} {'cos stock':>9s}")
for n, numel, e, c, es, cs in r["rows"]:
    print(f"{n:44s} {numel:9d} {e:9.4f} {es:9.4f} {c:9.4f} {cs:9.4f}")
