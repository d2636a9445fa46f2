"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerLightMamba2Net` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerLightMamba2Net.py:18-156) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerLightMamba2Net, nnUNetTrainerLightMamba2NetP  # noqa: F401

__all__ = ['nnUNetTrainerLightMamba2Net', 'nnUNetTrainerLightMamba2NetP']
