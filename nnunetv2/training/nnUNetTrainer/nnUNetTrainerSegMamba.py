"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerSegMamba` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerSegMamba.py:15-110) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerSegMamba  # noqa: F401

__all__ = ['nnUNetTrainerSegMamba']
