"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerMambaND2Net` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerMambaND2Net.py:15-156) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerMambaND2Net, nnUNetTrainerMambaND2NetP  # noqa: F401

__all__ = ['nnUNetTrainerMambaND2Net', 'nnUNetTrainerMambaND2NetP']
