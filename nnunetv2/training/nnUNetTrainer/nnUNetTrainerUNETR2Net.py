"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerUNETR2Net` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerUNETR2Net.py:15-118) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerUNETR2Net  # noqa: F401

__all__ = ['nnUNetTrainerUNETR2Net']
