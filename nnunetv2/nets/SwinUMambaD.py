"""`nnunetv2.nets.SwinUMambaD` of the reference (/root/reference/nnunetv2/nets/SwinUMambaD.py:22-732) -> native implementation in `nnuzoo_amd.nets.swin_umamba`."""
from nnuzoo_amd.nets.swin_umamba import FinalPatchExpand_X4, PatchEmbed2D, PatchExpand, PatchMerging2D, SS2D, SwinUMambaD, UNetResDecoder, VSSBlock, VSSLayer, VSSMEncoder, get_swin_umamba_d_from_plans, load_pretrained_ckpt  # noqa: F401

__all__ = ['FinalPatchExpand_X4', 'PatchEmbed2D', 'PatchExpand', 'PatchMerging2D', 'SS2D', 'SwinUMambaD', 'UNetResDecoder', 'VSSBlock', 'VSSLayer', 'VSSMEncoder', 'get_swin_umamba_d_from_plans', 'load_pretrained_ckpt']
