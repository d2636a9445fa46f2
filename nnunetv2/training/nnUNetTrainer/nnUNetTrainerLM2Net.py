"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerLM2Net` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerLM2Net.py:16-158) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerLM2Net, nnUNetTrainerLM2NetP  # noqa: F401

__all__ = ['nnUNetTrainerLM2Net', 'nnUNetTrainerLM2NetP']
