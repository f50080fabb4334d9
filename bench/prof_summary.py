"""Per-step kernel summary of a rocprofv3 kernel trace (rocpd SQLite) of bench.py.

    python bench/prof_summary.py gpurun_out/prof/x_results.db [steps_to_average]
"""
import sys

sys.path.insert(0, ".")
from distributed_deep_learning_amd.utils.profiling import format_summary, summarize_kernel_trace  # noqa: E402

print(format_summary(summarize_kernel_trace(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5)))
