"""`nnunetv2.nets.swt` of the reference (/root/reference/nnunetv2/nets/swt.py) -> native implementation in `nnuzoo_amd.nets.swt`."""
from nnuzoo_amd.nets.swt import BasicBlock, BasicBlockUp, DropPath, FinalPatchExpanding, Mlp, PatchEmbedding, PatchExpanding, PatchMerging, SwinTransformerBlock, SwinTransformerUnet, WindowAttention, get_dwconv_layer, get_swin_transformer_unet  # noqa: F401

__all__ = ['SwinTransformerUnet', 'DropPath', 'PatchEmbedding', 'PatchMerging', 'PatchExpanding', 'FinalPatchExpanding', 'Mlp', 'WindowAttention', 'SwinTransformerBlock', 'BasicBlock', 'BasicBlockUp', 'get_swin_transformer_unet']
