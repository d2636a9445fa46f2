"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerSSND2Net` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerSSND2Net.py:18-142) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerSSND2Net, nnUNetTrainerSSND2NetP  # noqa: F401

__all__ = ['nnUNetTrainerSSND2Net', 'nnUNetTrainerSSND2NetP']
