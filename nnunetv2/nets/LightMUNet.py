"""`nnunetv2.nets.LightMUNet` of the reference (/root/reference/nnunetv2/nets/LightMUNet.py) -> native implementation in `nnuzoo_amd.nets.lightmunet`."""
from nnuzoo_amd.nets.light_mamba2net import GSC, InstanceNorm, ResUpBlock  # noqa: F401
from nnuzoo_amd.nets.lightmunet import LightMUNet, MambaLayer, ResMambaBlock, get_dwconv_layer, get_from_plans, get_mamba_layer  # noqa: F401

__all__ = ['GSC', 'InstanceNorm', 'LightMUNet', 'MambaLayer', 'ResMambaBlock', 'ResUpBlock', 'get_dwconv_layer', 'get_from_plans', 'get_mamba_layer']
