"""`nnunetv2.training.loss.compound_losses` of the reference (/root/reference/nnunetv2/training/loss/compound_losses.py:8-105) -> native implementation in `nnuzoo_amd.training.loss`."""
from nnuzoo_amd.training.loss import DC_and_CE_loss, DC_and_BCE_loss  # noqa: F401

__all__ = ['DC_and_CE_loss', 'DC_and_BCE_loss']
