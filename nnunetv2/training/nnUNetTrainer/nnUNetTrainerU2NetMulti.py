"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerU2NetMulti` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerU2NetMulti.py:14-194) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerU2NetMulti, nnUNetTrainerU2NetMultiP  # noqa: F401

__all__ = ['nnUNetTrainerU2NetMulti', 'nnUNetTrainerU2NetMultiP']
