"""MambaND2Net ("MambaND" X^2-Net of the zoo) - reference: /root/reference/nnunetv2/nets/mamba_nd2net.py
  Block :565-666, create_block :672-722, MambaNDCore :725-1001, MambaND :1055-1297, MambaND2Net :1598-1905,
  get_mamband2net_from_plans :1907-1934; trainer nnUNetTrainerMambaND2Net.py.

Every stage of the outer U^2 is a UNETR-shaped inner net (`MambaND`) whose "transformer" is a stack of 1-D Mamba blocks
run over the patch tokens in alternating orderings: layers 2i, 2i+1 use ordering i mod n_orders of ('t h w', 't w h'
[, 'w h t' in 3-D]) and every odd layer walks its sequence backwards (:844, :973-994).  The mixer is `MambaSSM`
(nnuzoo_amd.nets.mamba_simple: in/x/dt/out projections + the HIP causal-conv1d / selective-scan / gate kernels of
csrc/mamba_block.hip, selective_scan.hip); the token re-orderings are views + one copy.  Patch embedding = depthwise
conv (kernel = stride = patch) + pointwise conv (:171-188), encoder / decoder around it = the monai UNETR blocks
restated in nets/monai_blocks.py (PARITY UNPINNED, see there); the outer wiring, patch merging / expansion and the
side outputs follow the reference class line by line (same attribute names -> same state_dict keys).

Reference quirks kept (observable behaviour): `Block` in the non-fused form used here adds the mixer output to the
NORMED input, not to the block input (:640-646); MambaNDCore returns every layer's output and MambaND taps layers
`linspace(2, num_layers - 1, 3)`; in 3-D the token grid handed to the blocks is (D', H', H') (:966-968); stage 5 -> 6
has no down-sampling (patch_merging5 / patch_expand5d use scale (1, 1, 1)).
"""
from __future__ import annotations

import itertools
from functools import partial

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..layer_norm import LayerNorm
from ..utilities.network_initialization import InitWeights_He
from .common2d import Convolution
from .mamba_simple import MambaSSM
from .monai_blocks import UnetOutBlock, UnetrBasicBlock, UnetrPrUpBlock, UnetrUpBlock
from .ssnd2net import PatchExpand as _PatchExpandND
from .ssnd2net import PatchMerging2D as _PatchMergingND
from .ssnd2net import _heads, _upsample_like, get_scale_value, get_scales, permute, shape


class DropPath(nn.Module):
    """floor(keep + U[0,1)) / keep per sample (:510-547)"""

    def __init__(self, drop_prob: float = 0.1):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        rt = keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)
        return x.div(keep) * rt.floor()


class Dropout(nn.Dropout):
    def __init__(self, drop_prob: float = 0.5, inplace: bool = False):
        super().__init__(p=drop_prob, inplace=inplace)


def get_dwconv_layer(spatial_dims, in_channels, out_channels, kernel_size=3, stride=1, bias=False, padding=None):
    """depthwise conv (kernel, stride) + pointwise conv of stride 1 (:171-188; the swt2net.py variant strides both)"""
    depth = Convolution(spatial_dims, in_channels, in_channels, strides=stride, kernel_size=kernel_size, bias=bias,
                        conv_only=True, groups=in_channels, padding=padding)
    point = Convolution(spatial_dims, in_channels, out_channels, strides=1, kernel_size=1, bias=bias, conv_only=True,
                        groups=1, padding=padding)
    return nn.Sequential(depth, point)


class PatchEmbed(nn.Module):
    """conv patch embedding -> (B, tokens, embed_dims) and the first two grid sizes (:189-312; tuple padding: no
    adaptive padding is built)"""

    def __init__(self, spatial_dims, in_channels, embed_dims, kernel_size, stride, padding, dilation, bias, input_size):
        super().__init__()
        padding, dilation = tuple(padding[:spatial_dims]), tuple(dilation[:spatial_dims])
        self.embed_dims = embed_dims
        self.adaptive_padding = None
        self.projection = get_dwconv_layer(spatial_dims, in_channels, embed_dims, kernel_size=tuple(kernel_size),
                                           stride=tuple(stride), padding=padding, bias=bias)
        self.norm = None
        self.init_input_size = tuple(input_size)
        self.init_out_size = tuple((input_size[a] + 2 * padding[a] - dilation[a] * (kernel_size[a] - 1) - 1) // stride[a] + 1
                                   for a in range(2))

    def forward(self, x):
        x = self.projection(x)
        out_size = (x.shape[2], x.shape[3])
        return x.flatten(2).transpose(1, 2), out_size


_ORDER_PERM = {"t h w": None, "t w h": (0, 1, 3, 2, 4), "w h t": (0, 3, 2, 1, 4)}   # each is its own inverse


class Block(nn.Module):
    def __init__(self, spatial_dims, dim, mixer_cls, norm_cls=LayerNorm, reverse=False, drop_path_rate=0.0,
                 drop_rate=0.0):
        super().__init__()
        self.spatial_dims = spatial_dims
        self.residual_in_fp32, self.fused_add_norm = True, False
        self.mixer = mixer_cls(dim)
        self.norm = norm_cls(dim)
        self.reverse = reverse
        self.drop_path = DropPath(drop_prob=drop_path_rate)
        self.dropout = Dropout(drop_prob=drop_rate)
        self.ffn = None

    def forward(self, hidden_states, order="t h w", shape=None, skip=True):
        if self.spatial_dims == 3:
            t, h, w = shape
        else:
            (h, w), t = shape, 1
        B, T, C = hidden_states.shape
        perm = _ORDER_PERM[order]
        if perm is not None:
            hidden_states = hidden_states.view(B, t, h, w, C).permute(*perm).reshape(B, T, C)
        if self.reverse:
            hidden_states = hidden_states.flip(1)
        hidden_states = self.norm(hidden_states)
        mixed = self.drop_path(self.dropout(self.mixer(hidden_states)))
        hidden_states = hidden_states + mixed if skip else mixed
        if self.reverse:
            hidden_states = hidden_states.flip(1)
        if perm is not None:
            dims = [t, h, w]
            pdims = [dims[p - 1] for p in perm[1:4]]
            hidden_states = hidden_states.view(B, *pdims, C).permute(*perm).reshape(B, T, C)
        return hidden_states


def create_block(spatial_dims, d_model, ssm_cfg=None, norm_epsilon=1e-5, layer_idx=None, reverse=None, drop_rate=0.1,
                 drop_path_rate=0.1):
    mixer_cls = partial(MambaSSM, layer_idx=layer_idx, **(ssm_cfg or {}))
    block = Block(spatial_dims, d_model, mixer_cls, norm_cls=partial(LayerNorm, eps=norm_epsilon), reverse=reverse,
                  drop_rate=drop_rate, drop_path_rate=drop_path_rate)
    block.layer_idx = layer_idx
    return block


class MambaNDCore(nn.Module):
    def __init__(self, spatial_dims, img_size, patch_size, in_channels, embed_dims, num_layers, drop_rate=0.,
                 drop_path_rate=0., d_state=16):
        super().__init__()
        self.spatial_dims, self.embed_dims, self.img_size, self.num_layers = spatial_dims, embed_dims, img_size, num_layers
        self.n_dim_pos = 4
        self.patch_embed = PatchEmbed(spatial_dims, in_channels, embed_dims, kernel_size=patch_size, stride=patch_size,
                                      padding=(0, 0, 0), dilation=(1, 1, 1), bias=True, input_size=img_size)
        pr = self.patch_embed.init_out_size
        self.patch_resolution = (pr[0], pr[1], pr[1])
        self.drop_after_pos = nn.Dropout(p=drop_rate)
        dpr = np.linspace(0, drop_path_rate, num_layers)
        self.layers = nn.ModuleList([
            create_block(spatial_dims, embed_dims, ssm_cfg={"d_state": d_state}, drop_rate=drop_rate,
                         drop_path_rate=float(dpr[i]), reverse=(i % 2) > 0) for i in range(num_layers)])
        self.pre_norm = nn.Identity()
        self.final_norm = False
        self.ln1 = nn.Identity()

    def forward(self, x):
        x, pr = self.patch_embed(x)
        if self.spatial_dims == 3:
            orders, grid = ("t h w", "t w h", "w h t"), (pr[0], pr[1], pr[1])
        else:
            orders, grid = ("t h w", "t w h"), (pr[0], pr[1])
        x = self.pre_norm(self.drop_after_pos(x))
        outs = []
        for i, blk in enumerate(self.layers):
            x = blk(x, order=orders[(i // 2) % len(orders)], shape=grid)
            outs.append(x)
        return outs[-1], outs


class MambaND(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, img_size, feature_size=16, hidden_size=768,
                 norm_name="instance", conv_block=False, res_block=True, dropout_rate=0.0, num_layers=7,
                 patch_size=(16, 16, 16), decoder_scale=(2, 2, 2, 2), encoder_scale=(2, 2, 2), encoder_layers=(2, 1, 0)):
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise AssertionError("dropout_rate should be between 0 and 1.")
        sd = self.spatial_dims = spatial_dims
        self.feature_size, self.hidden_size, self.img_size = feature_size, hidden_size, img_size
        self.patch_size = tuple(patch_size[:sd])
        self.feat_size = tuple(int(img_size[a] // self.patch_size[a]) for a in range(sd))
        self.classification = False
        self.out_indices = [int(v) for v in np.linspace(2, num_layers - 1, 3)]
        self.mamba = MambaNDCore(sd, img_size, self.patch_size, in_channels, hidden_size, num_layers,
                                 drop_rate=dropout_rate, drop_path_rate=dropout_rate)
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
        x, hidden = self.mamba(x_in)
        enc1 = self.encoder1(x_in)
        enc2 = self.encoder2(self.proj_feat(hidden[self.out_indices[0]]))
        enc3 = self.encoder3(self.proj_feat(hidden[self.out_indices[1]]))
        enc4 = self.encoder4(self.proj_feat(hidden[self.out_indices[2]]))
        dec3 = self.decoder5(self.proj_feat(x), enc4)
        dec2 = self.decoder4(dec3, enc3)
        dec1 = self.decoder3(dec2, enc2)
        return self.out(self.decoder2(dec1, enc1))


class PatchMerging2D(_PatchMergingND):
    """this file's variant of the N-D patch merge (:1382-1473): `scale[:spatial_dims]`, and ONE part order for 2-D and
    3-D - itertools.product over the axes, later axes fastest (ssnd2net.py's 2-D branch uses (0,0),(1,0),(0,1),(1,1))"""

    def __init__(self, spatial_dims, input_dim, scale, output_features, norm_layer=LayerNorm):
        super().__init__(spatial_dims, input_dim, scale if isinstance(scale, int) else tuple(scale[:spatial_dims]),
                         output_features, norm_layer)

    def forward(self, x, permute_=False):
        sd = self.spatial_dims
        if permute_:
            x = permute(x, sd).contiguous()
        B, Z, H, W, C = shape(x, sd, channel_first=False)
        if (H % self.hs == 1) or (W % self.ws == 1) or (Z and (Z % self.zs == 1)):
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2, 0, Z % 2)) if sd == 3 else F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        B, Z, H, W, C = shape(x, sd, channel_first=False)
        if sd == 3:
            rng = [[0] if s == 1 else [0, 1] for s in (self.zs, self.hs, self.ws)]
            parts = [x[:, c[0]::self.zs, c[1]::self.hs, c[2]::self.ws, :] for c in itertools.product(*rng)]
            out_shape = (B, Z // self.zs, H // self.hs, W // self.ws, (self.zs * self.hs * self.ws) * C)
        else:
            rng = [[0] if s == 1 else [0, 1] for s in (self.hs, self.ws)]
            parts = [x[:, c[0]::self.hs, c[1]::self.ws, :] for c in itertools.product(*rng)]
            out_shape = (B, H // self.hs, W // self.ws, (self.ws * self.hs) * C)
        x = torch.cat([t for t in parts if t.numel() != 0], -1).view(out_shape)
        x = self.reduction(self.norm(x))
        if permute_:
            x = permute(x, sd, reverse=True).contiguous()
        return x


class PatchExpand(_PatchExpandND):
    def __init__(self, spatial_dims, dim, scale, output_dim=None, norm_layer=LayerNorm):
        super().__init__(spatial_dims=spatial_dims, dim=dim, scale=scale if isinstance(scale, int) else
                         tuple(scale[:spatial_dims]), output_dim=output_dim, norm_layer=norm_layer)


class _UnetrStageX2(nn.Module):
    """outer U^2 wiring shared by MambaND2Net and UNETR2Net (the two reference constructors are identical up to the
    inner stage class: nets/mamba_nd2net.py:1598-1809, nets/unetr2net.py:1026-1240)"""

    def _build(self, stage_cls, spatial_dims: int, in_ch: int, out_ch: int, deep_supervision: bool, input_patch_size):
        sd = self.spatial_dims = spatial_dims
        self.deep_supervision, self.input_patch_size = deep_supervision, input_patch_size
        self.scales = scales = get_scales(sd, input_patch_size, n_layers=5, patch_size=None)

        def ips(k):
            return input_patch_size if k == 0 else get_scale_value(sd, input_patch_size, scales[:k])

        mnd = partial(stage_cls, spatial_dims=sd)
        deep = dict(encoder_layers=(0, 0, 0), decoder_scale=(2, 1, 1, 1))
        self.stage1 = mnd(in_channels=in_ch, out_channels=32, feature_size=4, hidden_size=96, num_layers=7,
                          patch_size=(16, 16, 16), img_size=ips(0))
        self.patch_merging1 = PatchMerging2D(sd, 32, scale=scales[0], output_features=64)
        self.stage2 = mnd(in_channels=64, out_channels=64, feature_size=4, hidden_size=192, num_layers=6,
                          patch_size=(16, 16, 16), img_size=ips(1))
        self.patch_merging2 = PatchMerging2D(sd, 64, scale=scales[1], output_features=128)
        self.stage3 = mnd(in_channels=128, out_channels=128, feature_size=8, hidden_size=384, num_layers=5,
                          patch_size=(8, 8, 8), img_size=ips(2), decoder_scale=(2, 2, 2, 1))
        self.patch_merging3 = PatchMerging2D(sd, 128, scale=scales[2], output_features=256)
        self.stage4 = mnd(in_channels=256, out_channels=256, feature_size=8, hidden_size=384, num_layers=4,
                          patch_size=(4, 4, 4), img_size=ips(3), encoder_layers=(1, 1, 0), decoder_scale=(2, 2, 1, 1))
        self.patch_merging4 = PatchMerging2D(sd, 256, scale=scales[3], output_features=512)
        self.stage5 = mnd(in_channels=512, out_channels=512, feature_size=16, hidden_size=384, num_layers=4,
                          patch_size=(2, 2, 2), img_size=ips(4), **deep)
        self.patch_merging5 = PatchMerging2D(sd, 512, scale=(1, 1, 1), output_features=512)
        self.stage6 = mnd(in_channels=512, out_channels=512, feature_size=16, hidden_size=384, num_layers=4,
                          patch_size=(2, 2, 2), img_size=ips(4), **deep)
        # decoder
        self.patch_expand5d = PatchExpand(sd, dim=512, scale=(1, 1, 1), norm_layer=LayerNorm, output_dim=512)
        self.stage5d = mnd(in_channels=1024, out_channels=512, feature_size=16, hidden_size=384, num_layers=4,
                           patch_size=(2, 2, 2), img_size=ips(4), **deep)
        self.patch_expand4d = PatchExpand(sd, dim=512, scale=scales[-2], norm_layer=LayerNorm, output_dim=256)
        self.concat_back_dim4d = nn.Linear(512, 256)
        self.stage4d = mnd(in_channels=256, out_channels=256, feature_size=8, hidden_size=384, num_layers=4,
                           patch_size=(2, 2, 2), img_size=ips(3), **deep)
        self.patch_expand3d = PatchExpand(sd, dim=256, scale=scales[-3], norm_layer=LayerNorm, output_dim=128)
        self.concat_back_dim3d = nn.Linear(256, 128)
        self.stage3d = mnd(in_channels=128, out_channels=128, feature_size=4, hidden_size=384, num_layers=5,
                           patch_size=(4, 4, 4), img_size=ips(2), encoder_layers=(1, 1, 0), decoder_scale=(2, 2, 1, 1))
        self.patch_expand2d = PatchExpand(sd, dim=128, scale=scales[-4], norm_layer=LayerNorm, output_dim=64)
        self.concat_back_dim2d = nn.Linear(128, 64)
        self.stage2d = mnd(in_channels=64, out_channels=64, feature_size=4, hidden_size=192, num_layers=6,
                           patch_size=(8, 8, 8), img_size=ips(1), decoder_scale=(2, 2, 2, 1))
        self.patch_expand1d = PatchExpand(sd, dim=64, scale=scales[-5], norm_layer=LayerNorm, output_dim=32)
        self.concat_back_dim1d = nn.Linear(64, 32)
        self.stage1d = mnd(in_channels=32, out_channels=32, feature_size=4, hidden_size=96, num_layers=7,
                           patch_size=(16, 16, 16), img_size=ips(0))
        for i, c in enumerate([32, 64, 128, 256, 512, 512]):
            setattr(self, f"side{i + 1}", Convolution(sd, c, out_ch, kernel_size=1, padding=0, conv_only=True))
        self.outconv = Convolution(sd, 6 * out_ch, out_ch, kernel_size=1, conv_only=True)

    def forward(self, x):
        sd = self.spatial_dims
        hx1 = self.stage1(x)
        hx2 = self.stage2(self.patch_merging1(hx1, permute_=True))
        hx3 = self.stage3(self.patch_merging2(hx2, permute_=True))
        hx4 = self.stage4(self.patch_merging3(hx3, permute_=True))
        hx5 = self.stage5(self.patch_merging4(hx4, permute_=True))
        hx6 = self.stage6(self.patch_merging5(hx5, permute_=True))
        hx5d = self.stage5d(torch.cat((self.patch_expand5d(hx6, permute_=True), hx5), 1))
        d, dec = hx5d, {5: hx5d}
        for lvl, skip in ((4, hx4), (3, hx3), (2, hx2), (1, hx1)):
            up = getattr(self, f"patch_expand{lvl}d")(d)                                        # channel last
            up = getattr(self, f"concat_back_dim{lvl}d")(torch.cat((up, permute(skip, sd)), -1))
            d = getattr(self, f"stage{lvl}d")(permute(up, sd, reverse=True))
            dec[lvl] = d
        sides = [self.side1(dec[1]), self.side2(dec[2]), self.side3(dec[3]), self.side4(dec[4]), self.side5(dec[5]),
                 self.side6(hx6)]
        d0 = self.outconv(torch.cat([sides[0]] + [_upsample_like(s, sides[0]) for s in sides[1:]], 1))
        return (d0, *sides) if self.deep_supervision else d0

    def _encoder_groups(self):
        return [getattr(self, f"stage{i}") for i in range(1, 7)] + [getattr(self, f"patch_merging{i}") for i in range(1, 5)]

    @torch.no_grad()
    def freeze_encoder(self):
        for g in self._encoder_groups():   # (the reference additionally names a non-existent `pool56`: it raises there)
            for p in g.parameters():
                p.requires_grad = False

    @torch.no_grad()
    def unfreeze_encoder(self):
        for g in self._encoder_groups():
            for p in g.parameters():
                p.requires_grad = True


class MambaND2Net(_UnetrStageX2):
    def __init__(self, spatial_dims: int, in_ch: int, out_ch: int, deep_supervision: bool, input_patch_size):
        super().__init__()
        self._build(MambaND, spatial_dims, in_ch, out_ch, deep_supervision, input_patch_size)


def get_mamband2net_from_plans(plans_manager, dataset_json: dict, configuration_manager, num_input_channels: int,
                               deep_supervision: bool = True, use_pretrain: bool = True, small_mode: bool = False):
    if small_mode:
        raise NotImplementedError()                 # as the reference (:1926)
    model = MambaND2Net(spatial_dims=len(configuration_manager.patch_size), in_ch=num_input_channels,
                        out_ch=_heads(plans_manager, dataset_json), deep_supervision=deep_supervision,
                        input_patch_size=configuration_manager.patch_size)
    model.apply(InitWeights_He(1e-2))
    return model
