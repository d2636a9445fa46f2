"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainer` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:67-1700) -> native implementation in `nnuzoo_amd.training.nnUNetTrainer`."""
from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer  # noqa: F401

__all__ = ['nnUNetTrainer']
