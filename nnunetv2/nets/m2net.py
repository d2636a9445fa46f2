"""`nnunetv2.nets.m2net` of the reference (/root/reference/nnunetv2/nets/m2net.py) -> native implementation in `nnuzoo_amd.nets.m2net`."""
from nnuzoo_amd.nets.m2net import SS2D, VSSBlock, VSSLayer, PatchEmbed2D, VSSMEncoder, VSSMDecoder, MU, M2Net, M2NetP, get_m2net_from_plans, get_m2netp_from_plans  # noqa: F401
from nnuzoo_amd.nets.common2d import REBNCONV, RSU4F, PatchMerging2D, PatchExpand, get_dwconv_layer, _upsample_like  # noqa: F401

__all__ = ['SS2D', 'VSSBlock', 'VSSLayer', 'PatchEmbed2D', 'VSSMEncoder', 'VSSMDecoder', 'MU', 'M2Net', 'M2NetP', 'get_m2net_from_plans', 'get_m2netp_from_plans']
