"""`nnunetv2.nets.u2net_multi` of the reference (/root/reference/nnunetv2/nets/u2net_multi.py) -> native implementation in `nnuzoo_amd.nets.u2net_multi`."""
from nnuzoo_amd.nets.u2net_multi import MaxPool, RSU7, RSU6, RSU5, RSU4, RSU4F, U2NET, U2NETP, _upsample_like, get_u2net_from_plans, get_u2netp_from_plans  # noqa: F401

__all__ = ['MaxPool', 'RSU7', 'RSU6', 'RSU5', 'RSU4', 'RSU4F', 'U2NET', 'U2NETP', 'get_u2net_from_plans', 'get_u2netp_from_plans']
