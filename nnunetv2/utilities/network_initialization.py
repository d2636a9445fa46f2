"""`nnunetv2.utilities.network_initialization` of the reference (/root/reference/nnunetv2/utilities/network_initialization.py:4-12) -> native implementation in `nnuzoo_amd.utilities.network_initialization`."""
from nnuzoo_amd.utilities.network_initialization import InitWeights_He  # noqa: F401

__all__ = ['InitWeights_He']
