"""`nnunetv2.nets.light_mamba2net` of the reference (/root/reference/nnunetv2/nets/light_mamba2net.py) -> native implementation in `nnuzoo_amd.nets.light_mamba2net`."""
from nnuzoo_amd.nets.light_mamba2net import GSC, InstanceNorm, LightMUNet, LightMamba2Net, LightMamba2NetP, MambaLayer, MaxPool, PatchExpand, PatchMerging2D, ResMambaBlock, ResUpBlock, get_dwconv_layer, get_light_mamba2net_from_plans, get_scale_value, get_scales  # noqa: F401

__all__ = ['GSC', 'InstanceNorm', 'LightMUNet', 'LightMamba2Net', 'LightMamba2NetP', 'MambaLayer', 'MaxPool', 'PatchExpand', 'PatchMerging2D', 'ResMambaBlock', 'ResUpBlock', 'get_dwconv_layer', 'get_light_mamba2net_from_plans', 'get_scale_value', 'get_scales']
