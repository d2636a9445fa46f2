"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerU2Net` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerU2Net.py:14-124) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerU2Net, nnUNetTrainerU2NetP  # noqa: F401

__all__ = ['nnUNetTrainerU2Net', 'nnUNetTrainerU2NetP']
