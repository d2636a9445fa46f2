"""`nnunetv2.nets.mamba_nd2net` of the reference (/root/reference/nnunetv2/nets/mamba_nd2net.py) -> native implementation in `nnuzoo_amd.nets.mamba_nd2net`."""
from nnuzoo_amd.nets.mamba_nd2net import Block, create_block, MambaNDCore, MambaND, MambaND2Net, PatchEmbed, PatchMerging2D, PatchExpand, get_dwconv_layer, get_mamband2net_from_plans  # noqa: F401

__all__ = ['Block', 'create_block', 'MambaNDCore', 'MambaND', 'MambaND2Net', 'PatchEmbed', 'PatchMerging2D', 'PatchExpand', 'get_dwconv_layer', 'get_mamband2net_from_plans']
