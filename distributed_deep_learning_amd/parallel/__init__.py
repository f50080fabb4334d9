"""Parallelism strategies: flat arena, DP (all-reduce and PS-style), inter-layer MP, micro-batch pipeline."""
from .arena import CHUNK, ParamArena, plan_buckets
from .ddp import DistributedDataParallel
from .mp import (FnStage, StagedModel, partition_layers, reference_partition_cnn, reference_partition_lstm,
                 reference_partition_mlp, stage_modules)
from .pipeline import PipelinedModel, schedule
from .ps import ShardedDataParallel

__all__ = ["CHUNK", "ParamArena", "plan_buckets", "DistributedDataParallel", "FnStage", "StagedModel",
           "partition_layers", "reference_partition_cnn", "reference_partition_lstm", "reference_partition_mlp",
           "stage_modules", "PipelinedModel", "schedule", "ShardedDataParallel"]
