"""`nnunetv2.nets.lm2net` of the reference (/root/reference/nnunetv2/nets/lm2net.py) -> native implementation in `nnuzoo_amd.nets.lm2net`."""
from nnuzoo_amd.nets.light_mamba2net import GSC, InstanceNorm, MaxPool, ResUpBlock  # noqa: F401
from nnuzoo_amd.nets.lm2net import LM2Net, LM2NetP, LightMUNet, MambaLayer, PatchExpand, PatchMerging2D, RSU4F, ResMambaBlock, get_lm2net_from_plans, get_scale_value, get_scales  # noqa: F401

__all__ = ['GSC', 'InstanceNorm', 'LM2Net', 'LM2NetP', 'LightMUNet', 'MambaLayer', 'MaxPool', 'PatchExpand', 'PatchMerging2D', 'RSU4F', 'ResMambaBlock', 'ResUpBlock', 'get_lm2net_from_plans', 'get_scale_value', 'get_scales']
