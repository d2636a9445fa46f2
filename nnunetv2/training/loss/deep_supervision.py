"""`nnunetv2.training.loss.deep_supervision` of the reference (/root/reference/nnunetv2/training/loss/deep_supervision.py:5-30) -> native implementation in `nnuzoo_amd.training.loss`."""
from nnuzoo_amd.training.loss import DeepSupervisionWrapper  # noqa: F401

__all__ = ['DeepSupervisionWrapper']
