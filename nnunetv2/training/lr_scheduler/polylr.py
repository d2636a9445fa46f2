"""`nnunetv2.training.lr_scheduler.polylr` of the reference (/root/reference/nnunetv2/training/lr_scheduler/polylr.py:7-26) -> native implementation in `nnuzoo_amd.training.lr_scheduler`."""
from nnuzoo_amd.training.lr_scheduler import PolyLRScheduler  # noqa: F401

__all__ = ['PolyLRScheduler']
