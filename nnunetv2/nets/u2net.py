"""`nnunetv2.nets.u2net` of the reference (/root/reference/nnunetv2/nets/u2net.py) -> native implementation in `nnuzoo_amd.nets.u2net`."""
from nnuzoo_amd.nets.u2net import REBNCONV, RSU7, RSU6, RSU5, RSU4, RSU4F, U2NET, U2NETP, _upsample_like, get_u2net_from_plans, get_u2netp_from_plans  # noqa: F401

__all__ = ['REBNCONV', 'RSU7', 'RSU6', 'RSU5', 'RSU4', 'RSU4F', 'U2NET', 'U2NETP', 'get_u2net_from_plans', 'get_u2netp_from_plans']
