"""`nnunetv2.training.nnUNetTrainer.nnUNetTrainerSwinTransformerUnet` of the reference (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerSwinTransformerUnet.py:17-108) -> native implementation in `nnuzoo_amd.training.zoo_trainers`."""
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerSwinTransformerUnet  # noqa: F401

__all__ = ['nnUNetTrainerSwinTransformerUnet']
