"""`nnunetv2.training.loss.dice` of the reference (/root/reference/nnunetv2/training/loss/dice.py:58-119) -> native implementation in `nnuzoo_amd.training.loss`."""
from nnuzoo_amd.training.loss import MemoryEfficientSoftDiceLoss  # noqa: F401

__all__ = ['MemoryEfficientSoftDiceLoss']
