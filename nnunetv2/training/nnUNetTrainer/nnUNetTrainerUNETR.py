"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerUNETR` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerUNETR.py:13-150) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerUNETR  # noqa: F401

__all__ = ['nnUNetTrainerUNETR']
