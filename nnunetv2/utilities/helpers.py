"""`nnunetv2.utilities.helpers` of the reference (/root/reference/nnunetv2/utilities/helpers.py:8-9) -> native implementation in `nnuzoo_amd.training.loss`."""
from nnuzoo_amd.training.loss import softmax_helper_dim1  # noqa: F401

__all__ = ['softmax_helper_dim1']
