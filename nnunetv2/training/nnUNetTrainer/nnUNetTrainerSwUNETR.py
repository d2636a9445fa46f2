"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerSwUNETR` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerSwUNETR.py:13-99) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerSwUNETR  # noqa: F401

__all__ = ['nnUNetTrainerSwUNETR']
