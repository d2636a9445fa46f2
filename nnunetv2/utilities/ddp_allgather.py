"""`nnunetv2.utilities.ddp_allgather` of the reference (/root/reference/nnunetv2/utilities/ddp_allgather.py:25-49) -> native implementation in `nnuzoo_amd.training.loss`."""
from nnuzoo_amd.training.loss import AllGatherGrad  # noqa: F401

__all__ = ['AllGatherGrad']
