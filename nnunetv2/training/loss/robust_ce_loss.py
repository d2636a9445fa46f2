"""`nnunetv2.training.loss.robust_ce_loss` of the reference (/root/reference/nnunetv2/training/loss/robust_ce_loss.py:6-16) -> native implementation in `nnuzoo_amd.training.loss`."""
from nnuzoo_amd.training.loss import RobustCrossEntropyLoss  # noqa: F401

__all__ = ['RobustCrossEntropyLoss']
