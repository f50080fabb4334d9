"""Child process of test_gpu_round2: every CONV_CASES shape (fwd / dgrad / wgrad) under whatever DDL_* environment the
parent set (the kernel-selection knobs are read once per process)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import test_gpu_kernels as T  # noqa: E402

failed = 0
for case in T.CONV_CASES:
    try:
        T.test_conv_fwd_dgrad_wgrad(case)
    except AssertionError as e:
        failed += 1
        print("FAIL", case, str(e)[:300], flush=True)
print(f"checked {len(T.CONV_CASES)} cases, {failed} failed", flush=True)
sys.exit(1 if failed else 0)
