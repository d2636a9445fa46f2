"""`nnunetv2.nets.LightSS2DMambaUNet` of the reference (/root/reference/nnunetv2/nets/LightSS2DMambaUNet.py:18-583) -> native implementation in `nnuzoo_amd.nets.light_ss2d_mamba_unet`."""
from nnuzoo_amd.nets.light_ss2d_mamba_unet import GSC, LightSS2DMambaUNet, MambaLayer, ResMambaBlock, ResUpBlock, SS2D, get_dwconv_layer, get_mamba_layer, get_mamband2net_from_plans  # noqa: F401

__all__ = ['GSC', 'LightSS2DMambaUNet', 'MambaLayer', 'ResMambaBlock', 'ResUpBlock', 'SS2D', 'get_dwconv_layer', 'get_mamba_layer', 'get_mamband2net_from_plans']
