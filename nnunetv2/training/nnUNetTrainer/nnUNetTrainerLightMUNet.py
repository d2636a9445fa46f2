"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerLightMUNet` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerLightMUNet.py:14-129) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerLightMUNet  # noqa: F401

__all__ = ['nnUNetTrainerLightMUNet']
