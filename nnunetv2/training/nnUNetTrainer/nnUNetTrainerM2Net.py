"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerM2Net` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerM2Net.py:15-130) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerM2Net, nnUNetTrainerM2NetP  # noqa: F401

__all__ = ['nnUNetTrainerM2Net', 'nnUNetTrainerM2NetP']
