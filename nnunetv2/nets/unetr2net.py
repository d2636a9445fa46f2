"""`nnunetv2.nets.unetr2net` of the reference (/root/reference/nnunetv2/nets/unetr2net.py) -> native implementation in `nnuzoo_amd.nets.unetr2net`."""
from nnuzoo_amd.nets.unetr2net import UNETR, UNETR2Net, ViT, get_unetr2net_from_plans  # noqa: F401
from nnuzoo_amd.nets.mamba_nd2net import PatchMerging2D, PatchExpand, get_dwconv_layer  # noqa: F401

__all__ = ['UNETR', 'UNETR2Net', 'ViT', 'get_unetr2net_from_plans']
