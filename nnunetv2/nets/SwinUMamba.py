"""`nnunetv2.nets.SwinUMamba` of the reference (/root/reference/nnunetv2/nets/SwinUMamba.py:21-683) -> native implementation in `nnuzoo_amd.nets.swin_umamba`."""
from nnuzoo_amd.nets.swin_umamba import PatchEmbed2D, PatchMerging2D, SS2D, SwinUMamba, VSSBlock, VSSLayer, VSSMEncoder, get_swin_umamba_from_plans, load_pretrained_ckpt  # noqa: F401

__all__ = ['PatchEmbed2D', 'PatchMerging2D', 'SS2D', 'SwinUMamba', 'VSSBlock', 'VSSLayer', 'VSSMEncoder', 'get_swin_umamba_from_plans', 'load_pretrained_ckpt']
