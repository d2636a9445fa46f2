"""`nets/swt.py` of the reference (/root/reference/nnunetv2/nets/swt.py:28-155 SwinTransformerUnet and its blocks; factory
`get_swin_transformer_unet` :505-531): the single Swin U-net.  Its classes are the ones the X^2 variant wraps - the
reference keeps byte-identical copies in swt2net.py - so the native implementations are shared (window attention on
csrc/window_attention.hip, fp32 Linear / Mlp layers on csrc/dense32.hip, LayerNorm on csrc/layer_norm.hip)."""
from __future__ import annotations

from functools import partial

from ..layer_norm import LayerNorm
from .swt2net import (BasicBlock, BasicBlockUp, DropPath, FinalPatchExpanding, Mlp, PatchEmbedding, PatchExpanding,  # noqa: F401
                      PatchMerging, SwinTransformerBlock, SwinTransformerUnet, WindowAttention, get_dwconv_layer)


def get_swin_transformer_unet(num_segmentation_heads: int, num_input_channels: int, deep_supervision: bool = True,
                              use_pretrain: bool = True):
    """swt.py:505-531: depths (2, 2, 9, 2), embed_dim 96, heads (3, 6, 12, 24), 7x7 windows, LayerNorm eps 1e-6; no
    deep supervision (a single-output net), weights from the class's own trunc-normal init"""
    return SwinTransformerUnet(patch_size=4, in_ch=num_input_channels, out_ch=num_segmentation_heads,
                               depths=(2, 2, 9, 2), embed_dim=96, num_heads=(3, 6, 12, 24), window_size=7, qkv_bias=True,
                               mlp_ratio=4, drop_path_rate=0.1, drop_rate=0, attn_drop_rate=0,
                               norm_layer=partial(LayerNorm, eps=1e-6))
