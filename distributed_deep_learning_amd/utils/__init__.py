"""Auxiliary subsystems: seeding, profiling (HIP-event timers, rocprofv3 helpers), debug checks, HIP-graph capture."""
from .debug import GradLedger, check_finite, replica_divergence
from .graph import GraphedEvalStep, GraphedTrainStep
from .host import freeze_gc, unfreeze_gc
from .profiling import RegionTimer, pmc_groups, rocprof_command, summarize_kernel_trace
from .seed import seed_everything

__all__ = ["GradLedger", "check_finite", "replica_divergence", "GraphedEvalStep", "GraphedTrainStep", "RegionTimer", "pmc_groups",
           "rocprof_command", "summarize_kernel_trace", "seed_everything", "freeze_gc", "unfreeze_gc"]
