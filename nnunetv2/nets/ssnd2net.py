"""`nnunetv2.nets.ssnd2net` of the reference (/root/reference/nnunetv2/nets/ssnd2net.py) -> native implementation in `nnuzoo_amd.nets.ssnd2net`."""
from nnuzoo_amd.nets.ssnd2net import PatchMerging2D, PatchExpand, PatchEmbed2D, InstanceNorm, GSC, VSSBlock, VSSLayer, VSSMEncoder, VSSMDecoder, MU, SSND2Net, SSND2NetP, get_ssnd2net_from_plans, get_m2net_from_plans, permute, shape, get_scale, get_scale_value, get_scales  # noqa: F401
from nnuzoo_amd.nets.ssnd import SSND  # noqa: F401

__all__ = ['PatchMerging2D', 'PatchExpand', 'PatchEmbed2D', 'InstanceNorm', 'GSC', 'VSSBlock', 'VSSLayer', 'VSSMEncoder', 'VSSMDecoder', 'MU', 'SSND2Net', 'SSND2NetP', 'get_ssnd2net_from_plans', 'get_m2net_from_plans', 'permute', 'shape', 'get_scale', 'get_scale_value', 'get_scales']
