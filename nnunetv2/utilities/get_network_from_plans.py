"""`nnunetv2.utilities.get_network_from_plans` of the reference (/root/reference/nnunetv2/utilities/get_network_from_plans.py:18-62) -> native implementation in `nnuzoo_amd.utilities.get_network_from_plans`."""
from nnuzoo_amd.utilities.get_network_from_plans import get_network_from_plans  # noqa: F401

__all__ = ['get_network_from_plans']
