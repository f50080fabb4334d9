from .fused import LARS, FusedAdam, FusedSGD, default_no_decay
from .lr_scheduler import StepLR, WarmupCosineLR, WarmupPolyLR

__all__ = ["FusedSGD", "LARS", "FusedAdam", "default_no_decay", "StepLR", "WarmupPolyLR", "WarmupCosineLR"]
