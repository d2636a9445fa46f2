"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerSwinUMambaD` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerSwinUMambaD.py:17-124) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerSwinUMambaD  # noqa: F401

__all__ = ['nnUNetTrainerSwinUMambaD']
