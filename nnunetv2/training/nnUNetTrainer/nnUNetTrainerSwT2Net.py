"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerSwT2Net` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerSwT2Net.py:15-131) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerSwT2Net  # noqa: F401

__all__ = ['nnUNetTrainerSwT2Net']
