"""UNETR2Net ("UNETR" X^2-Net of the zoo) - reference: /root/reference/nnunetv2/nets/unetr2net.py (UNETR2Net :1026-1343,
UNETR :1346-1563, get_unetr2net_from_plans :1566-1600; trainer nnUNetTrainerUNETR2Net.py).

The outer U^2 is the wiring of MambaND2Net (identical constructor, nets/mamba_nd2net.py:_UnetrStageX2); every stage is a
`UNETR`: monai's `ViT` (conv patch embedding + learnable position embedding, pre-norm transformer blocks with GLOBAL
multi-head self-attention over <= 1024 patch tokens, head_dim 8 / 16 / 32) tapped at layers linspace(2, L - 1, 3), the
monai UNETR encoder / decoder blocks, and a depthwise + pointwise residual conv of the input (`rebnconvin`, hard-wired 2-D
in the reference, :1399).  monai is absent here: `ViT`, `TransformerBlock`, `SABlock`, `MLPBlock`,
`PatchEmbeddingBlock` are restated below from the published monai 1.3 sources - PARITY UNPINNED (structure, registration
order and parameter names follow that release; nothing can be checked against it in this image).

The attention core softmax(q k^T / sqrt(d)) v runs through `global_attention` (nnuzoo_amd/global_attention.py).
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from ..global_attention import global_attention
from ..layer_norm import LayerNorm
from ..utilities.network_initialization import InitWeights_He
from .mamba_nd2net import _UnetrStageX2, get_dwconv_layer
from .monai_blocks import UnetOutBlock, UnetrBasicBlock, UnetrPrUpBlock, UnetrUpBlock
from .ssnd2net import _heads, permute


class MLPBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, dropout_rate=0.0):
        super().__init__()
        self.linear1 = nn.Linear(hidden_size, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)
        self.fn = nn.GELU()
        self.drop1 = nn.Dropout(dropout_rate)
        self.drop2 = nn.Dropout(dropout_rate)

    def forward(self, x):
        return self.drop2(self.linear2(self.drop1(self.fn(self.linear1(x)))))


class SABlock(nn.Module):
    def __init__(self, hidden_size, num_heads, dropout_rate=0.0, qkv_bias=False, save_attn=False):
        super().__init__()
        if hidden_size % num_heads != 0:
            raise ValueError("hidden size should be divisible by num_heads.")
        if save_attn:
            raise NotImplementedError("save_attn: the attention matrix is never materialised")
        self.num_heads = num_heads
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        self.qkv = nn.Linear(hidden_size, hidden_size * 3, bias=qkv_bias)
        self.drop_output = nn.Dropout(dropout_rate)
        self.drop_weights = nn.Dropout(dropout_rate)
        self.head_dim = hidden_size // num_heads
        self.scale = self.head_dim ** -0.5
        self.dropout_rate = dropout_rate

    def forward(self, x):
        B, L, _ = x.shape
        # "b h (qkv l d) -> qkv b l h d": per token the qkv row is [q | k | v], each [head][head_dim]
        qkv = self.qkv(x).view(B, L, 3, self.num_heads, self.head_dim)
        if self.dropout_rate > 0 and self.training:
            raise NotImplementedError("attention dropout > 0 is not used by the zoo (dropout_rate 0.0)")
        o = global_attention(qkv, self.scale, self)                 # (B, L, heads * head_dim)
        return self.drop_output(self.out_proj(o))


class TransformerBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, num_heads, dropout_rate=0.0, qkv_bias=False, save_attn=False):
        super().__init__()
        self.mlp = MLPBlock(hidden_size, mlp_dim, dropout_rate)
        self.norm1 = LayerNorm(hidden_size)
        self.attn = SABlock(hidden_size, num_heads, dropout_rate, qkv_bias, save_attn)
        self.norm2 = LayerNorm(hidden_size)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class PatchEmbeddingBlock(nn.Module):
    """proj_type "conv", learnable position embedding (trunc-normal std 0.02)"""

    def __init__(self, in_channels, img_size, patch_size, hidden_size, num_heads, dropout_rate=0.0, spatial_dims=3):
        super().__init__()
        conv = {2: nn.Conv2d, 3: nn.Conv3d}[spatial_dims]
        self.n_patches = int(np.prod([i // p for i, p in zip(img_size, patch_size)]))
        self.patch_embeddings = conv(in_channels, hidden_size, kernel_size=tuple(patch_size), stride=tuple(patch_size))
        self.position_embeddings = nn.Parameter(torch.zeros(1, self.n_patches, hidden_size))
        self.dropout = nn.Dropout(dropout_rate)
        nn.init.trunc_normal_(self.position_embeddings, mean=0.0, std=0.02, a=-2.0, b=2.0)

    def forward(self, x):
        x = self.patch_embeddings(x).flatten(2).transpose(-1, -2)
        return self.dropout(x + self.position_embeddings)


class ViT(nn.Module):
    def __init__(self, in_channels, img_size, patch_size, hidden_size=768, mlp_dim=3072, num_layers=12, num_heads=12,
                 proj_type="conv", classification=False, dropout_rate=0.0, spatial_dims=3, qkv_bias=False,
                 save_attn=False):
        super().__init__()
        if proj_type != "conv" or classification:
            raise NotImplementedError("ViT as used by UNETR: conv patch projection, no classification head")
        self.patch_embedding = PatchEmbeddingBlock(in_channels, img_size, patch_size, hidden_size, num_heads,
                                                   dropout_rate, spatial_dims)
        self.blocks = nn.ModuleList([TransformerBlock(hidden_size, mlp_dim, num_heads, dropout_rate, qkv_bias, save_attn)
                                     for _ in range(num_layers)])
        self.norm = LayerNorm(hidden_size)

    def forward(self, x):
        x = self.patch_embedding(x)
        hidden = []
        for blk in self.blocks:
            x = blk(x)
            hidden.append(x)
        return self.norm(x), hidden


class UNETR(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, img_size, feature_size=16, hidden_size=768, mlp_dim=3072,
                 num_heads=12, proj_type="conv", norm_name="instance", conv_block=True, res_block=True,
                 dropout_rate=0.0, qkv_bias=False, save_attn=False, num_layers=7, patch_size=(16, 16, 16),
                 decoder_scale=(2, 2, 2, 2), encoder_scale=(2, 2, 2), encoder_layers=(2, 1, 0), add_last=True,
                 out_indices=None):
        super().__init__()
        self.add_last = add_last
        if add_last:
            self.rebnconvin = get_dwconv_layer(2, in_channels, out_channels)       # 2-D in the reference whatever spatial_dims
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        sd = self.spatial_dims = spatial_dims
        self.num_layers, self.hidden_size = num_layers, hidden_size
        img_size = tuple(int(v) for v in (img_size if isinstance(img_size, (tuple, list)) else (img_size,) * sd))
        self.patch_size = tuple(patch_size[:sd])
        self.feat_size = tuple(int(img_size[a] // self.patch_size[a]) for a in range(sd))
        self.classification = False
        self.out_indices = [int(v) for v in (np.linspace(2, num_layers - 1, 3) if out_indices is None else out_indices)]
        self.vit = ViT(in_channels, img_size, self.patch_size, hidden_size, mlp_dim, num_layers, num_heads, proj_type,
                       False, dropout_rate, sd, qkv_bias, save_attn)
        f = feature_size
        self.encoder1 = UnetrBasicBlock(sd, in_channels, f, 3, 1, norm_name, res_block)
        self.encoder2 = UnetrPrUpBlock(sd, hidden_size, f * 2, encoder_layers[0], 3, 1, encoder_scale[2], norm_name,
                                       conv_block, res_block)
        self.encoder3 = UnetrPrUpBlock(sd, hidden_size, f * 4, encoder_layers[1], 3, 1, encoder_scale[1], norm_name,
                                       conv_block, res_block)
        self.encoder4 = UnetrPrUpBlock(sd, hidden_size, f * 8, encoder_layers[2], 3, 1, encoder_scale[0], norm_name,
                                       conv_block, res_block)
        self.decoder5 = UnetrUpBlock(sd, hidden_size, f * 8, 3, decoder_scale[0], norm_name, res_block)
        self.decoder4 = UnetrUpBlock(sd, f * 8, f * 4, 3, decoder_scale[1], norm_name, res_block)
        self.decoder3 = UnetrUpBlock(sd, f * 4, f * 2, 3, decoder_scale[2], norm_name, res_block)
        self.decoder2 = UnetrUpBlock(sd, f * 2, f, 3, decoder_scale[3], norm_name, res_block)
        self.out = UnetOutBlock(sd, f, out_channels)

    def proj_feat(self, x):
        x = x.view(x.size(0), *self.feat_size, self.hidden_size)
        return permute(x, self.spatial_dims, reverse=True).contiguous()

    def forward(self, x_in):
        last_add = self.rebnconvin(x_in) if self.add_last else None
        x, hidden = self.vit(x_in)
        enc1 = self.encoder1(x_in)
        enc2 = self.encoder2(self.proj_feat(hidden[self.out_indices[0]]))
        enc3 = self.encoder3(self.proj_feat(hidden[self.out_indices[1]]))
        enc4 = self.encoder4(self.proj_feat(hidden[self.out_indices[2]]))
        dec3 = self.decoder5(self.proj_feat(x), enc4)
        dec2 = self.decoder4(dec3, enc3)
        dec1 = self.decoder3(dec2, enc2)
        out = self.out(self.decoder2(dec1, enc1))
        return out + last_add if self.add_last else out


class MonaiUNETR(UNETR):
    """`monai.networks.nets.UNETR` as nnUNetTrainerUNETR instantiates it (/root/reference/nnunetv2/training/nnUNetTrainer/
    nnUNetTrainerUNETR.py:10, :43-58): the stage class above in its original form - 12 transformer layers, 16-voxel patches, skip
    features after layers 3, 6, 9, no residual input branch.  Same child names as monai's module (vit, encoder1..4, decoder5..2,
    out).  monai is absent here: PARITY UNPINNED (nets/monai_blocks.py header); the attention, MLP and Linear layers are the
    pinned ones of this file."""

    def __init__(self, in_channels, out_channels, img_size, feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12,
                 proj_type="conv", norm_name="instance", conv_block=True, res_block=True, dropout_rate=0.0, spatial_dims=3,
                 qkv_bias=False, save_attn=False):
        super().__init__(spatial_dims, in_channels, out_channels, img_size, feature_size=feature_size, hidden_size=hidden_size,
                         mlp_dim=mlp_dim, num_heads=num_heads, proj_type=proj_type, norm_name=norm_name, conv_block=conv_block,
                         res_block=res_block, dropout_rate=dropout_rate, qkv_bias=qkv_bias, save_attn=save_attn, num_layers=12,
                         patch_size=(16, 16, 16), add_last=False, out_indices=(3, 6, 9))


class UNETR2Net(_UnetrStageX2):
    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, deep_supervision: bool, input_patch_size,
                 add_last: bool = True):
        super().__init__()
        from functools import partial
        self._build(partial(UNETR, add_last=add_last), spatial_dims, in_channels, out_channels, deep_supervision,
                    input_patch_size)


def get_unetr2net_from_plans(plans_manager, dataset_json: dict, configuration_manager, num_input_channels: int,
                             deep_supervision: bool = True, use_pretrain: bool = True, small_mode: bool = False):
    if small_mode:
        raise NotImplementedError()                 # as the reference (:1586)
    model = UNETR2Net(spatial_dims=len(configuration_manager.patch_size), in_channels=num_input_channels,
                      out_channels=_heads(plans_manager, dataset_json), deep_supervision=deep_supervision,
                      input_patch_size=configuration_manager.patch_size)
    model.apply(InitWeights_He(1e-2))
    return model
