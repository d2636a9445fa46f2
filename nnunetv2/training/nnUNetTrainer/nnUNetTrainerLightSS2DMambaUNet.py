"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerLightSS2DMambaUNet` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerLightSS2DMambaUNet.py:17-140) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerLightSS2DMambaUNet  # noqa: F401

__all__ = ['nnUNetTrainerLightSS2DMambaUNet']
