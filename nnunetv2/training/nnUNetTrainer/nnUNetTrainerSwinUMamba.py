"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerSwinUMamba` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerSwinUMamba.py:18-113) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerSwinUMamba  # noqa: F401

__all__ = ['nnUNetTrainerSwinUMamba']
