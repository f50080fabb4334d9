"""Child process of test_gpu_round2: the fused-finalize checks with DDL_FUSED_FINALIZE=2 (always fuse; the knob is read
once per process and the default only fuses for tensors of at most 16 MiB), plus the fused residual / dense units."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import test_gpu_round2 as T  # noqa: E402

CONV = [(32, 64, 56, 256, 1), (16, 128, 28, 128, 3), (8, 256, 14, 1024, 1), (4, 64, 9, 72, 3), (64, 64, 56, 64, 3)]
BWD = [(32, 64, 56, True), (16, 128, 28, True), (16, 256, 14, False), (8, 512, 7, True), (8, 1024, 7, True), (4, 96, 9, True),
       (64, 256, 56, True)]
failed = 0
for fn, cases in ((T.test_conv_with_fused_bn_finalize_equals_separate_finalize_every_time, CONV),
                  (T.test_bn_backward_with_fused_finalize_matches_fp32_every_time, BWD),
                  (T.test_dense_block_on_one_feature_buffer_matches_fp32_concat_formulation, [(4, 64, 16, 32, 2)])):
    for case in cases:
        try:
            fn(*case)
        except AssertionError as e:
            failed += 1
            print("FAIL", fn.__name__, case, str(e)[:300], flush=True)
print(f"checked {len(CONV) + len(BWD) + 1} cases, {failed} failed", flush=True)
sys.exit(1 if failed else 0)
