"""SSND2Net / SSND2NetP - the N-D (2-D and 3-D) U^2 state-space networks of the zoo, for MI355X.

Same class names, constructor arguments, sub-module / parameter names and shapes (identical state_dict) as
/root/reference/nnunetv2/nets/ssnd2net.py: PatchMerging2D :321-403, PatchExpand :426-521, VSSMDecoder :524-661,
PatchEmbed2D :664-695, InstanceNorm :698-709, GSC :711-754, VSSBlock :757-786, VSSLayer :789-857, VSSMEncoder :860-997,
scale helpers :1000-1067, MU :1070-1140, SSND2Net :1143-1405, SSND2NetP :1446-1707, factories :1710-1775.
The hot operator is the SSND block (nets/ssnd.py: 2*spatial_dims-direction selective scan on the gfx950 scan kernel);
everything around it is layout plumbing and library ops (LayerNorm, GEMMs, depthwise / 1x1 convs, InstanceNorm).

Written table-driven where the reference spells the twelve MU stages out one by one; reference quirks kept on purpose:
`get_scale` keeps an axis un-pooled when its size is odd, PatchMerging pads odd axes by one, decoder stages reuse
`input_features_skip` of the LAST loop iteration for the final seg layer.
"""
from __future__ import annotations

import itertools
import math
from functools import partial
from typing import Callable, List, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.init import trunc_normal_

from ..token_linear import TokenLinear
from ..layer_norm import LayerNorm

from ..utilities.network_initialization import InitWeights_He
from .common2d import Convolution, DropPath, get_dwconv_layer, residual_drop_path
from .common2d import _upsample_like as _upsample_like_2d
from .ssnd import SSND


def permute(x, spatial_dims: int, reverse=False):
    """channel-first <-> channel-last"""
    if spatial_dims == 2:
        return x.permute(0, 3, 1, 2) if reverse else x.permute(0, 2, 3, 1)
    if spatial_dims == 3:
        return x.permute(0, 4, 1, 2, 3) if reverse else x.permute(0, 2, 3, 4, 1)
    raise ValueError


def shape(x, spatial_dims, channel_first=True):
    if spatial_dims == 2:
        if channel_first:
            B, C, H, W = x.shape
            return B, C, None, H, W
        B, H, W, C = x.shape
        return B, None, H, W, C
    if spatial_dims == 3:
        if channel_first:
            return tuple(x.shape)
        B, Z, H, W, C = x.shape
        return B, Z, H, W, C
    raise Exception()


def _upsample_like(src, tar, upsample_mode="nontrainable"):
    """monai UpSample(nontrainable, InterpolateMode.LINEAR, align_corners=False) to tar's spatial size"""
    if src.dim() == 4:
        return _upsample_like_2d(src, tar.shape[2:])        # torch's forward, matmul-adjoint backward (common2d)
    return F.interpolate(src, size=tar.shape[2:], mode="trilinear", align_corners=False)


def get_scale(scale_value, scale_factor=2):
    f = 1 if scale_value % scale_factor == 1 else scale_factor
    return f, scale_value // f


def get_scale_value(spatial_dims: int, input_patch_size, scales):
    v = list(input_patch_size)
    assert len(v) == spatial_dims
    for s in scales:
        v = [a / b for a, b in zip(v, s)]
    return tuple(v)


def get_scales(spatial_dims, input_patch_size, n_layers, patch_size=None, min_size: int = 1):
    """per level and axis: halve unless the size is odd or (light_mamba2net.py:562-600) the half would fall below
    `min_size`; min_size=1 is ssnd2net.py's / mamba_nd2net.py's form"""
    if input_patch_size is None:
        return None
    v = list(input_patch_size)
    if patch_size is not None:
        p = (patch_size,) * spatial_dims if isinstance(patch_size, int) else tuple(patch_size)
        v = [a / b for a, b in zip(v, p)]
    scales = []
    for _ in range(n_layers):
        step = []
        for a in range(spatial_dims):
            f, half = get_scale(v[a])
            if half >= min_size:
                v[a] = half
            else:
                f = 1
            step.append(f)
        scales.append(tuple(step))
    return scales


class PatchMerging2D(nn.Module):
    """space-to-depth by `scale` per axis, LayerNorm, Linear to output_features (N-D despite the name)"""

    def __init__(self, spatial_dims: int, input_dim: int, scale, output_features: int, norm_layer=LayerNorm):
        super().__init__()
        self.spatial_dims = spatial_dims
        if spatial_dims == 2:
            self.hs, self.ws = (scale, scale) if isinstance(scale, int) else scale
            self.zs = 1
        else:
            self.zs, self.hs, self.ws = (scale, scale, scale) if isinstance(scale, int) else scale
        self.input_feature_size = (self.zs * self.ws * self.hs) * input_dim
        self.output_features = output_features
        self.reduction = TokenLinear(self.input_feature_size, self.output_features, bias=False)
        self.norm = norm_layer(self.input_feature_size)

    def forward(self, x, permute_=False):
        sd = self.spatial_dims
        if permute_:
            x = permute(x, sd).contiguous()
        B, Z, H, W, C = shape(x, sd, channel_first=False)
        if (H % self.hs == 1) or (W % self.ws == 1) or (Z and (Z % self.zs == 1)):
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2, 0, Z % 2)) if sd == 3 else F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        B, Z, H, W, C = shape(x, sd, channel_first=False)
        if sd == 2:
            parts = [x[:, 0::self.hs, 0::self.ws, :], x[:, 1::self.hs, 0::self.ws, :],
                     x[:, 0::self.hs, 1::self.ws, :], x[:, 1::self.hs, 1::self.ws, :]]
            x = torch.cat([t for t in parts if t.numel() != 0], -1)
            x = x.view(B, H // self.hs, W // self.ws, (self.ws * self.hs) * C)
        else:
            rng = [[0] if s == 1 else [0, 1] for s in (self.zs, self.hs, self.ws)]
            parts = [x[:, c[0]::self.zs, c[1]::self.hs, c[2]::self.ws, :] for c in itertools.product(*rng)]
            x = torch.cat([t for t in parts if t.numel() != 0], -1)
            x = x.view(B, Z // self.zs, H // self.hs, W // self.ws, (self.zs * self.hs * self.ws) * C)
        x = self.reduction(self.norm(x))
        if permute_:
            x = permute(x, sd, reverse=True).contiguous()
        return x


class PatchExpand(nn.Module):
    """depth-to-space by `scale` per axis around a Linear + LayerNorm (N-D)"""

    def __init__(self, spatial_dims: int, dim: int, scale, output_dim: int = None, norm_layer=LayerNorm):
        super().__init__()
        self.spatial_dims, self.dim, self.output_dim = spatial_dims, dim, output_dim
        if isinstance(scale, int):
            self.zs = self.hs = self.ws = self.cs = scale
            self.zs = self.zs if spatial_dims == 3 else 1
        else:
            if len(scale) == 4 and spatial_dims == 3:
                self.zs, self.hs, self.ws, self.cs = scale
            elif len(scale) == 3 and spatial_dims == 3:
                (self.zs, self.hs, self.ws), self.cs = scale, None
            elif len(scale) == 3 and spatial_dims == 2:
                self.hs, self.ws, self.cs = scale
                self.zs = 1
            elif len(scale) == 2 and spatial_dims == 2:
                (self.hs, self.ws), self.cs, self.zs = scale, None, 1
            else:
                raise Exception()
            if self.output_dim is not None and self.cs is not None:
                raise ValueError("output_dim and cs cannot be not None at the same time!")
        if self.output_dim is None:
            self.expand = TokenLinear(dim, self.zs * self.hs * self.ws // self.cs, bias=False)
            self.norm = norm_layer(dim // self.cs)
        else:
            self.expand = TokenLinear(dim // (self.zs * self.hs * self.ws), self.output_dim, bias=False)
            self.norm = norm_layer(self.output_dim)

    def _d2s(self, x):
        """(B, [Z,] H, W, p*c) -> (B, [Z*zs,] H*hs, W*ws, c)"""
        if self.spatial_dims == 2:
            B, H, W, C = x.shape
            c = C // (self.hs * self.ws)
            return x.view(B, H, W, self.hs, self.ws, c).permute(0, 1, 3, 2, 4, 5).reshape(B, H * self.hs, W * self.ws, c)
        B, Z, H, W, C = x.shape
        c = C // (self.zs * self.hs * self.ws)
        return x.view(B, Z, H, W, self.zs, self.hs, self.ws, c).permute(0, 1, 4, 2, 5, 3, 6, 7) \
            .reshape(B, Z * self.zs, H * self.hs, W * self.ws, c)

    def forward(self, x, permute_=False):
        x = permute(x, self.spatial_dims)  # channel last
        if self.output_dim is None:
            x = self._d2s(self.expand(x))
        else:
            x = self.expand(self._d2s(x))
        x = self.norm(x)
        if permute_:
            x = permute(x, self.spatial_dims, reverse=True).contiguous()
        return x


class PatchEmbed2D(nn.Module):
    def __init__(self, spatial_dims: int = 2, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None, **kwargs):
        super().__init__()
        if isinstance(patch_size, int):
            patch_size = (patch_size,) * spatial_dims
        self.spatial_dims = spatial_dims
        self.proj = Convolution(spatial_dims, in_chans, embed_dim, kernel_size=patch_size, strides=patch_size,
                                conv_only=True)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        x = permute(self.proj(x), self.spatial_dims)
        return self.norm(x) if self.norm is not None else x


class InstanceNorm(nn.Module):
    def __init__(self, spatial_dims: int, in_channels: int):
        super().__init__()
        self.layer = (nn.InstanceNorm2d if spatial_dims == 2 else nn.InstanceNorm3d)(in_channels)

    def forward(self, input):
        return self.layer(input)


class GSC(nn.Module):
    """gated spatial convolution in front of every VSS block: two normalised conv branches, a third on their sum"""

    def __init__(self, spatial_dims: int, in_channels) -> None:
        super().__init__()
        self.proj = get_dwconv_layer(spatial_dims, in_channels=in_channels, out_channels=in_channels, stride=1, bias=True)
        self.norm = InstanceNorm(spatial_dims, in_channels)
        self.nonliner = nn.ReLU()
        self.proj2 = Convolution(spatial_dims, in_channels, in_channels, kernel_size=1, strides=1, padding=0,
                                 conv_only=True)
        self.norm2 = InstanceNorm(spatial_dims, in_channels)
        self.nonliner2 = nn.ReLU()
        self.proj3 = get_dwconv_layer(spatial_dims, in_channels=in_channels, out_channels=in_channels, stride=1, bias=True)
        self.norm3 = InstanceNorm(spatial_dims, in_channels)
        self.nonliner3 = nn.ReLU()

    def forward(self, x):
        x1 = self.nonliner(self.proj(self.norm(x)))
        x2 = self.nonliner2(self.proj2(self.norm2(x)))
        s = x1 + x2
        return self.nonliner3(self.proj3(self.norm3(s))) + x


class VSSBlock(nn.Module):
    def __init__(self, spatial_dims: int, factorization_type: str, hidden_dim: int = 0, drop_path: float = 0,
                 norm_layer: Callable[..., torch.nn.Module] = partial(LayerNorm, eps=1e-6), attn_drop_rate: float = 0,
                 d_state: int = 16, dilation: int = 1, **kwargs):
        super().__init__()
        self.spatial_dims = spatial_dims
        self.ln_1 = norm_layer(hidden_dim)
        if isinstance(self.ln_1, LayerNorm):
            self.ln_1.feeds_linear = True     # its only consumer is SSND.in_proj
        self.gsc = GSC(spatial_dims=spatial_dims, in_channels=hidden_dim)
        self.self_attention = SSND(spatial_dims=spatial_dims, factorization_type=factorization_type, d_model=hidden_dim,
                                   dropout=attn_drop_rate, d_state=d_state, dilation=dilation, **kwargs)
        self.drop_path = DropPath(drop_path)

    def forward(self, input: torch.Tensor):
        input = permute(self.gsc(permute(input, self.spatial_dims, reverse=True)), self.spatial_dims)
        return residual_drop_path(input, self.self_attention(self.ln_1(input)), self.drop_path)


class VSSLayer(nn.Module):
    def __init__(self, spatial_dims: int, factorization_type: str, dim, depth, attn_drop=0., drop_path=0.,
                 norm_layer=LayerNorm, downsample=None, use_checkpoint=False, d_state=16, dilation: int = 1):
        super().__init__()
        self.dim, self.use_checkpoint = dim, use_checkpoint
        self.blocks = nn.ModuleList([
            VSSBlock(spatial_dims=spatial_dims, factorization_type=factorization_type, hidden_dim=dim,
                     drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer,
                     attn_drop_rate=attn_drop, d_state=d_state, dilation=dilation) for i in range(depth)])

        def _init_weights(module: nn.Module):
            # the reference draws (and discards) a kaiming init for every out_proj here: only the RNG stream moves
            for name, p in module.named_parameters():
                if name in ["out_proj.weight"]:
                    nn.init.kaiming_uniform_(p.clone().detach_(), a=math.sqrt(5))

        self.apply(_init_weights)
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return self.downsample(x) if self.downsample is not None else x


def _vssm_init(m: nn.Module):
    if isinstance(m, nn.Linear):
        trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


class VSSMEncoder(nn.Module):
    def __init__(self, spatial_dims: int, factorization_type: str, patch_size=4, in_chans=3, depths=[2, 2, 9, 2],
                 dims=[96, 192, 384, 768], d_state=16, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1,
                 norm_layer=LayerNorm, patch_norm=True, use_checkpoint=False, add_last: bool = False,
                 out_ch: int = None, scales: List[Tuple[int, ...]] = None, dilations: list = None):
        super().__init__()
        self.scales, self.spatial_dims, self.num_layers, self.add_last = scales, spatial_dims, len(depths), add_last
        if isinstance(dims, int):
            dims = [int(dims * 2 ** i) for i in range(self.num_layers)]
        if self.add_last:
            self.rebnconvin = get_dwconv_layer(spatial_dims, in_chans, out_ch)
        self.embed_dim, self.dims = dims[0], dims
        self.patch_embed = PatchEmbed2D(spatial_dims=spatial_dims, patch_size=patch_size,
                                        in_chans=out_ch if self.add_last else in_chans, embed_dim=self.embed_dim,
                                        norm_layer=norm_layer if patch_norm else None)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        self.downsamples = nn.ModuleList()
        dilations = dilations or [1] * self.num_layers
        for i in range(self.num_layers):
            self.layers.append(VSSLayer(
                spatial_dims=spatial_dims, factorization_type=factorization_type, dim=dims[i], depth=depths[i],
                d_state=math.ceil(dims[0] / 6) if d_state is None else d_state, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer, downsample=None,
                use_checkpoint=use_checkpoint, dilation=dilations[i]))
            if i < self.num_layers - 1 and scales is not None and np.prod(scales[i]) != 1:
                self.downsamples.append(PatchMerging2D(spatial_dims=spatial_dims, input_dim=dims[i], scale=scales[i],
                                                       output_features=dims[i + 1], norm_layer=norm_layer))
        self.apply(_vssm_init)

    def forward(self, x):
        x_ret = []
        if self.add_last:
            x = self.rebnconvin(x)
            x_ret.append(x)
        else:
            x_ret.append(None)
        x = self.pos_drop(self.patch_embed(x))
        for s, layer in enumerate(self.layers):
            x = layer(x)
            x_ret.append(permute(x, self.spatial_dims, reverse=True))
            if s < len(self.downsamples):
                x = self.downsamples[s](x)
        return x_ret


class VSSMDecoder(nn.Module):
    def __init__(self, spatial_dims: int, factorization_type: str, num_classes: int, deep_supervision,
                 features_per_stage=None, depths=None, drop_path_rate: float = 0.2, d_state: int = 16,
                 patch_size: int = 4, scales=None, dilations: int = None):
        super().__init__()
        self.spatial_dims, self.deep_supervision, self.num_classes, self.depths = spatial_dims, deep_supervision, \
            num_classes, depths
        enc = features_per_stage
        n = len(enc)
        dpr = [x.item() for x in torch.linspace(drop_path_rate, 0, (n - 1) * 2)]
        stages, expand_layers, seg_layers, concat_back_dim = [], [], [], []
        dilations = dilations or [1] * (n - 1)
        skip = 0
        for s in range(1, n):
            below, skip = enc[-s], enc[-(s + 1)]
            if scales is not None and np.prod(scales[-s]) != 1:
                expand_layers.append(PatchExpand(spatial_dims=spatial_dims, dim=below, scale=scales[-s], output_dim=below,
                                                 norm_layer=LayerNorm))
            else:
                expand_layers.append(None)
            stages.append(VSSLayer(spatial_dims=spatial_dims, factorization_type=factorization_type, dim=skip, depth=1,
                                   attn_drop=0., drop_path=dpr[sum(depths[:s - 1]):sum(depths[:s])],
                                   d_state=math.ceil(2 * skip / 6) if d_state is None else d_state,
                                   norm_layer=LayerNorm, downsample=None, use_checkpoint=False,
                                   dilation=dilations[s - 1]))
            seg_layers.append(Convolution(spatial_dims, skip, num_classes, 1, 1, padding=0, bias=True, conv_only=True))
            concat_back_dim.append(TokenLinear(2 * skip, skip))
        if patch_size != 1:
            expand_layers.append(PatchExpand(spatial_dims=spatial_dims, dim=enc[0], scale=patch_size,
                                             norm_layer=LayerNorm))
        else:
            expand_layers.append(None)
        stages.append(nn.Identity())
        seg_layers.append(Convolution(spatial_dims, skip, num_classes, 1, 1, padding=0, bias=True, conv_only=True))
        self.stages = nn.ModuleList(stages)
        self.expand_layers = nn.ModuleList(expand_layers)
        self.seg_layers = nn.ModuleList(seg_layers)
        self.concat_back_dim = nn.ModuleList(concat_back_dim)

    def forward(self, skips):
        sd = self.spatial_dims
        lres = skips[-1]
        outs = []
        last = len(self.stages) - 1
        for s in range(len(self.stages)):
            x = permute(lres, sd) if self.expand_layers[s] is None else self.expand_layers[s](lres)
            if s < last:
                x = self.concat_back_dim[s](torch.cat((x, permute(skips[-(s + 2)], sd)), -1))
            x = permute(self.stages[s](x), sd, reverse=True)
            if self.deep_supervision:
                outs.append(self.seg_layers[s](x))
            elif s == last:
                outs.append(self.seg_layers[-1](x))
            lres = x
        outs = outs[::-1]
        return outs if self.deep_supervision else outs[0]


class MU(nn.Module):
    """one stage of the outer U^2: a small state-space U-Net (VSSM encoder + decoder) with a residual input conv"""

    def __init__(self, spatial_dims: int, factorization_type: str, in_ch: int, mid_ch, out_ch: int, n_layers: int,
                 patch_size=4, add_last: bool = False, input_patch_size=None, vss_args: dict = None,
                 decoder_args: dict = None):
        super().__init__()
        self.add_last, self.input_patch_size = add_last, input_patch_size
        features, depths = [mid_ch] * n_layers, [2] * n_layers
        self.scales = scales = get_scales(spatial_dims, input_patch_size, n_layers - 1, patch_size)
        self.vssm_encoder = VSSMEncoder(**dict(
            spatial_dims=spatial_dims, factorization_type=factorization_type, in_chans=in_ch, patch_size=patch_size,
            depths=depths, dims=features, add_last=add_last, out_ch=out_ch if add_last else None, scales=scales,
            drop_path_rate=0.2, **(vss_args or dict())))
        self.vssm_decoder = VSSMDecoder(**dict(
            spatial_dims=spatial_dims, factorization_type=factorization_type, num_classes=out_ch, deep_supervision=False,
            features_per_stage=features, drop_path_rate=0.2, d_state=16, depths=depths, scales=scales,
            patch_size=patch_size, **(decoder_args or dict())))

    def forward(self, x):
        skips = self.vssm_encoder(x)
        out = self.vssm_decoder(skips)
        return out + skips[0] if self.add_last else out

    @torch.no_grad()
    def freeze_encoder(self):
        for name, param in self.vssm_encoder.named_parameters():
            if "patch_embed" not in name:
                param.requires_grad = False

    @torch.no_grad()
    def unfreeze_encoder(self):
        for param in self.vssm_encoder.parameters():
            param.requires_grad = True


class _SSNDU2(nn.Module):
    """shared wiring of SSND2Net / SSND2NetP: six encoder MU stages with N-D patch merging between them, five decoder
    MU stages fed by patch expansion + skip concat (+ Linear), six side outputs fused by a 1x1 conv.
    cfg: enc = [(in, mid, out, n_layers)] x 6, merge_out = [c] x 5, dec = [(in, mid, out, n_layers)] x 5 (5d..1d),
         expand_out = [c] x 5 (5d..1d), concat_lin = [(in, out)] x 4 (4d..1d), side_in = [c] x 6"""

    def _build(self, spatial_dims, factorization_type, in_ch, out_ch, deep_supervision, input_patch_size, cfg):
        self.spatial_dims, self.deep_supervision, self.input_patch_size = spatial_dims, deep_supervision, input_patch_size
        self.scales = scales = get_scales(spatial_dims, input_patch_size, n_layers=5, patch_size=None)
        mu = partial(MU, spatial_dims=spatial_dims, factorization_type=factorization_type, patch_size=1, add_last=True)

        def ips(k):  # the reference passes the down-scaled patch size to stages 1-4 (and their decoders) only
            return input_patch_size if k == 0 else get_scale_value(spatial_dims, input_patch_size, scales[:k])

        enc = cfg["enc"]
        for i, (ci, cm, co, nl) in enumerate(enc):
            c_in = in_ch if i == 0 else ci
            extra = dict(input_patch_size=ips(i)) if i < 4 else {}
            setattr(self, f"stage{i + 1}", mu(in_ch=c_in, mid_ch=cm, out_ch=co, n_layers=nl, **extra))
            if i < 5:
                setattr(self, f"patch_merging{i + 1}", PatchMerging2D(spatial_dims, co, scale=scales[i],
                                                                      output_features=cfg["merge_out"][i]))
        # decoder, deepest first (names 5d, 4d, ... 1d)
        for j, (ci, cm, co, nl) in enumerate(cfg["dec"]):
            lvl = 5 - j                                   # 5, 4, 3, 2, 1
            dim_below = enc[5][2] if j == 0 else cfg["dec"][j - 1][2]
            setattr(self, f"patch_expand{lvl}d", PatchExpand(spatial_dims=spatial_dims, dim=dim_below, scale=scales[-(j + 1)],
                                                             norm_layer=LayerNorm, output_dim=cfg["expand_out"][j]))
            if j > 0:
                a, b = cfg["concat_lin"][j - 1]
                setattr(self, f"concat_back_dim{lvl}d", TokenLinear(a, b))
            extra = dict(input_patch_size=ips(lvl - 1)) if lvl <= 4 else {}
            setattr(self, f"stage{lvl}d", mu(in_ch=ci, mid_ch=cm, out_ch=co, n_layers=nl, **extra))
        for i, c in enumerate(cfg["side_in"]):
            setattr(self, f"side{i + 1}", Convolution(spatial_dims, c, out_ch, kernel_size=3, padding=1, conv_only=True))
        self.outconv = Convolution(spatial_dims, 6 * out_ch, out_ch, kernel_size=1, conv_only=True)

    def forward(self, x):
        sd = self.spatial_dims
        hx, enc = x, []
        for i in range(1, 7):
            h = getattr(self, f"stage{i}")(hx)
            enc.append(h)
            if i < 6:
                hx = getattr(self, f"patch_merging{i}")(h, permute_=True)
        hx6 = enc[5]
        up = self.patch_expand5d(hx6, permute_=True)
        d = self.stage5d(torch.cat((up, enc[4]), 1))
        dec = {5: d}
        for lvl in (4, 3, 2, 1):
            up = getattr(self, f"patch_expand{lvl}d")(d)                                    # channel last
            up = getattr(self, f"concat_back_dim{lvl}d")(torch.cat((up, permute(enc[lvl - 1], sd)), -1))
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
        for g in self._encoder_groups():
            for p in g.parameters():
                p.requires_grad = False

    @torch.no_grad()
    def unfreeze_encoder(self):
        for g in self._encoder_groups():
            for p in g.parameters():
                p.requires_grad = True


class SSND2Net(_SSNDU2):
    def __init__(self, spatial_dims: int, factorization_type: str, in_ch: int, out_ch: int, deep_supervision: bool,
                 input_patch_size):
        super().__init__()
        self._build(spatial_dims, factorization_type, in_ch, out_ch, deep_supervision, input_patch_size, dict(
            enc=[(None, 16, 32, 7), (64, 32, 64, 6), (128, 64, 128, 5), (256, 128, 256, 4), (512, 256, 512, 4),
                 (512, 256, 512, 4)],
            merge_out=[64, 128, 256, 512, 512],
            dec=[(1024, 256, 512, 4), (256, 128, 256, 4), (128, 64, 128, 5), (64, 32, 64, 6), (32, 16, 32, 7)],
            expand_out=[512, 256, 128, 64, 32],
            concat_lin=[(512, 256), (256, 128), (128, 64), (64, 32)],
            side_in=[32, 64, 128, 256, 512, 512]))


class SSND2NetP(_SSNDU2):
    def __init__(self, spatial_dims: int, factorization_type: str, in_ch: int, out_ch: int, deep_supervision: bool,
                 input_patch_size):
        super().__init__()
        self._build(spatial_dims, factorization_type, in_ch, out_ch, deep_supervision, input_patch_size, dict(
            enc=[(None, 16, 64, 7), (64, 16, 64, 6), (64, 16, 64, 5), (64, 16, 64, 4), (64, 16, 64, 4), (64, 16, 64, 4)],
            merge_out=[64, 64, 64, 64, 64],
            dec=[(128, 16, 128, 4), (128, 16, 128, 4), (128, 16, 128, 5), (128, 16, 128, 6), (128, 16, 128, 7)],
            expand_out=[64, 64, 64, 64, 64],
            concat_lin=[(128, 128), (128, 128), (128, 128), (128, 128)],
            side_in=[128, 128, 128, 128, 128, 64]))


def _heads(plans_manager, dataset_json) -> int:
    if plans_manager is not None and hasattr(plans_manager, "get_label_manager"):
        return plans_manager.get_label_manager(dataset_json).num_segmentation_heads
    return len(dataset_json["labels"])


def get_ssnd2net_from_plans(plans_manager, dataset_json: dict, configuration_manager, num_input_channels: int,
                            deep_supervision: bool = True, use_pretrain: bool = True, small_mode: bool = False):
    cls = SSND2NetP if small_mode else SSND2Net
    model = cls(spatial_dims=len(configuration_manager.patch_size), factorization_type="cross-scan",
                in_ch=num_input_channels, out_ch=_heads(plans_manager, dataset_json), deep_supervision=deep_supervision,
                input_patch_size=configuration_manager.patch_size)
    model.apply(InitWeights_He(1e-2))
    return model


def get_m2net_from_plans(plans_manager, dataset_json: dict, configuration_manager, num_input_channels: int,
                         deep_supervision: bool = True, use_pretrain: bool = True):
    """(the reference's ssnd2net.py keeps this name for the non-small SSND2Net factory)"""
    return get_ssnd2net_from_plans(plans_manager, dataset_json, configuration_manager, num_input_channels,
                                   deep_supervision, use_pretrain, small_mode=False)
