"""LightMamba2Net ("AltM^2-Net" / Alt1DM of the zoo) - reference: /root/reference/nnunetv2/nets/light_mamba2net.py
  MambaLayer :51-89, GSC :196-237, ResUpBlock :424-465, ResMambaBlock :468-537, get_scales :562-600, LightMUNet :605-781,
  LightMamba2Net :784-1008, LightMamba2NetP :1011-1275, get_light_mamba2net_from_plans :1279-1308;
  trainer nnUNetTrainerLightMamba2Net.py.

Every stage of the outer U^2 is a LightMUNet: a constant-width (mid_ch) residual U-Net whose encoder blocks are
ResMambaBlocks - a gated spatial convolution (GSC) followed by two 1-D Mamba2 layers that walk the voxels in the block's
axis order ('h w' / 'w h' in 2-D; 'd h w' / 'd w h' / 'w h d' in 3-D, cycling with the level).  Down-sampling is max
pooling by the per-axis scales of get_scales(min_size=4), up-sampling a 1x1 conv + (bi/tri)linear interpolation.

The mixer is nnuzoo_amd.nets.mamba2.Mamba2 (the SSD recurrence on the HIP chunk-scan / causal-conv1d / gate kernels);
LayerNorm is the HIP layer_norm kernel; everything else is small-channel fp32 convolution / GroupNorm / InstanceNorm glue.
Attribute names follow the reference class line by line (same state_dict keys).  monai helpers restated here:
get_upsample_layer(nontrainable) = nn.Upsample(linear, align_corners=False); get_norm_layer(("GROUP", {num_groups: 8}))
= nn.GroupNorm(8, C); get_act_layer(("RELU", {inplace})) = nn.ReLU (PARITY UNPINNED for those, as nets/monai_blocks.py).

Reference quirks kept: MambaLayer applies the SAME LayerNorm before the mixer and after the residual (:83-86); ResUpBlock's
docstring default act is SiLU but LightMUNet passes its own act (RELU); `add_last` is accepted and ignored (:769-771).
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from ..layer_norm import LayerNorm
from ..token_linear import TokenLinear
from ..utilities.network_initialization import InitWeights_He
from .common2d import Convolution, get_dwconv_layer
from .mamba2 import Mamba2
from .mamba_nd2net import PatchExpand, PatchMerging2D, _UnetrStageX2
from .ssnd2net import _heads, get_scale_value, get_scales

_GROUP8 = ("GROUP", {"num_groups": 8})
_RELU = ("RELU", {"inplace": True})


def get_norm_layer(name, spatial_dims, channels):
    kind, kw = (name, {}) if isinstance(name, str) else (name[0], dict(name[1]))
    if kind.lower() != "group":
        raise NotImplementedError(f"norm {name!r}: LightMUNet only builds GROUP norms")
    return nn.GroupNorm(num_channels=channels, **kw)


def get_act_layer(name):
    kind, kw = (name, {}) if isinstance(name, str) else (name[0], dict(name[1]))
    return {"relu": nn.ReLU, "silu": nn.SiLU}[kind.lower()](**kw)


def get_upsample_layer(spatial_dims, in_channels, upsample_mode="nontrainable", scale_factor=2):
    if str(getattr(upsample_mode, "value", upsample_mode)) != "nontrainable":
        raise NotImplementedError("LightMUNet is built with the non-trainable up-sampling only")
    sf = tuple(float(s) for s in scale_factor) if isinstance(scale_factor, (tuple, list)) else float(scale_factor)
    return nn.Upsample(scale_factor=sf, mode={2: "bilinear", 3: "trilinear"}[spatial_dims], align_corners=False)


class MambaLayer(nn.Module):
    @staticmethod
    def get_nheaddim(d_model, expand):
        """largest i < d_inner / 8 with (d_inner / i) % 8 == 0 (:53-58)"""
        nheaddim = 1
        for i in range(1, int(d_model * expand / 8)):
            if d_model * expand / i % 8 == 0:
                nheaddim = i
        return nheaddim

    #: the sequence mixer: Mamba2 here; nets/lm2net.py (the 1-D Mamba variant of the same file family) overrides it
    mixer = "mamba2"

    def __init__(self, input_dim, output_dim, d_state=16, d_conv=4, expand=2):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.norm = LayerNorm(input_dim)
        if self.mixer == "mamba2":
            self.mamba = Mamba2(d_model=input_dim, d_state=d_state, d_conv=d_conv, expand=expand,
                                headdim=self.get_nheaddim(input_dim, expand))
        else:
            from .mamba_simple import Mamba
            self.mamba = Mamba(d_model=input_dim, d_state=d_state, d_conv=d_conv, expand=expand)
        self.proj = TokenLinear(input_dim, output_dim)
        self.skip_scale = nn.Parameter(torch.ones(1))

    def forward(self, x):
        if x.dtype == torch.float16:
            x = x.type(torch.float32)
        B, C = x.shape[:2]
        assert C == self.input_dim
        img_dims = x.shape[2:]
        x_flat = x.reshape(B, C, -1).transpose(-1, -2)
        x_mamba = self.mamba(self.norm(x_flat)) + self.skip_scale * x_flat
        x_mamba = self.proj(self.norm(x_mamba))
        return x_mamba.transpose(-1, -2).reshape(B, self.output_dim, *img_dims)


class MaxPool(nn.Module):
    def __init__(self, spatial_dims: int, kernel_size, stride=None):
        super().__init__()
        self.max_pool = {2: nn.MaxPool2d, 3: nn.MaxPool3d}[spatial_dims](kernel_size=kernel_size, stride=stride)

    def forward(self, input):
        return self.max_pool(input)


class InstanceNorm(nn.Module):
    def __init__(self, spatial_dims: int, in_channels: int):
        super().__init__()
        self.layer = {2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}[spatial_dims](in_channels)

    def forward(self, input):
        return self.layer(input)


class GSC(nn.Module):
    """two parallel norm -> conv -> relu branches (depthwise-separable 3x3 and 1x1), summed, a third on the sum, residual"""

    def __init__(self, spatial_dims: int, in_channels) -> None:
        super().__init__()
        self.proj = get_dwconv_layer(spatial_dims, in_channels, in_channels, stride=1, bias=True)
        self.norm = InstanceNorm(spatial_dims, in_channels)
        self.nonliner = nn.ReLU()
        self.proj2 = Convolution(spatial_dims, in_channels, in_channels, kernel_size=1, strides=1, padding=0, conv_only=True)
        self.norm2 = InstanceNorm(spatial_dims, in_channels)
        self.nonliner2 = nn.ReLU()
        self.proj3 = get_dwconv_layer(spatial_dims, in_channels, in_channels, stride=1, bias=True)
        self.norm3 = InstanceNorm(spatial_dims, in_channels)
        self.nonliner3 = nn.ReLU()

    def forward(self, x):
        x1 = self.nonliner(self.proj(self.norm(x)))
        x2 = self.nonliner2(self.proj2(self.norm2(x)))
        x3 = self.nonliner3(self.proj3(self.norm3(x1 + x2)))
        return x3 + x


class ResUpBlock(nn.Module):
    def __init__(self, spatial_dims: int, in_channels: int, norm, kernel_size: int = 3, act=("SiLU", {"inplace": True})):
        super().__init__()
        if kernel_size % 2 != 1:
            raise AssertionError("kernel_size should be an odd number.")
        self.norm1 = get_norm_layer(norm, spatial_dims, in_channels)
        self.norm2 = get_norm_layer(norm, spatial_dims, in_channels)
        self.act = get_act_layer(act)
        self.conv = get_dwconv_layer(spatial_dims, in_channels, in_channels, kernel_size=kernel_size)
        self.skip_scale = nn.Parameter(torch.ones(1))

    def forward(self, x):
        identity = x
        x = self.act(self.norm1(x))
        x = self.conv(x) + self.skip_scale * identity
        return self.act(self.norm2(x))


_AXES = {2: "hw", 3: "dhw"}


class ResMambaBlock(nn.Module):
    layer_cls = MambaLayer

    def __init__(self, spatial_dims: int, in_channels: int, norm, kernel_size: int = 3, act=_RELU, order: str = "d h w"):
        super().__init__()
        if kernel_size % 2 != 1:
            raise AssertionError("kernel_size should be an odd number.")
        self.order, self.spatial_dims = order, spatial_dims
        self.gsc = GSC(spatial_dims, in_channels)
        self.norm1 = get_norm_layer(norm, spatial_dims, in_channels)
        self.norm2 = get_norm_layer(norm, spatial_dims, in_channels)
        self.act = get_act_layer(act)
        self.mamba1 = self.layer_cls(input_dim=in_channels, output_dim=in_channels)
        self.mamba2 = self.layer_cls(input_dim=in_channels, output_dim=in_channels)

    def nd_mamba_order(self, order: str, x: torch.Tensor, mamba_module: nn.Module):
        """runs the layer with the spatial axes permuted to `order` (the token sequence walks the LAST named axis
        fastest) and permutes back (:520-537)"""
        axes = _AXES[self.spatial_dims]
        perm = [0, 1] + [2 + axes.index(a) for a in order.split()]
        if perm == list(range(x.dim())):
            return mamba_module(x)
        inv = [perm.index(i) for i in range(x.dim())]
        return mamba_module(x.permute(perm)).permute(inv)

    def forward(self, x):
        x = self.gsc(x)
        identity = x
        x = self.nd_mamba_order(self.order, self.act(self.norm1(x)), self.mamba1)
        x = self.nd_mamba_order(self.order, self.act(self.norm2(x)), self.mamba2)
        return x + identity


class LightMUNet(nn.Module):
    block_cls = ResMambaBlock
    #: lm2net.py's LightMUNet (:223-402) differs in two points: `add_last` is honoured - a depthwise-separable conv of the
    #: input, created FIRST, is added to the output - and every encoder level has ONE ResMambaBlock (here: 1, 2, 2, ...)
    lm_variant = False

    def __init__(self, spatial_dims: int = 3, mid_ch: int = 32, in_ch: int = 1, out_ch: int = 2, dropout_prob=None,
                 act=_RELU, norm=_GROUP8, norm_name: str = "", num_groups: int = 8, use_conv_final: bool = True,
                 n_layers: int = 7, add_last: bool = False, upsample_mode="nontrainable", min_size: int = 4,
                 input_patch_size=None):
        super().__init__()
        if spatial_dims not in (2, 3):
            raise ValueError("`spatial_dims` can only be 2 or 3.")
        self.input_path_size, self.add_last, self.spatial_dims = input_patch_size, add_last, spatial_dims
        if self.lm_variant and add_last:
            self.rebnconvin = get_dwconv_layer(2, in_ch, out_ch)
        self.init_filters, self.in_channels, self.n_layers = mid_ch, in_ch, n_layers
        self.layer_in_channels = [mid_ch] * n_layers
        self.blocks_down = [1] + [1 if self.lm_variant else 2] * (n_layers - 1)
        self.blocks_up = [1] * (n_layers - 1)
        self.dropout_prob, self.act = dropout_prob, act
        self.act_mod = get_act_layer(act)
        self.scales = [(1, 1, 1)[:spatial_dims]] + get_scales(spatial_dims, input_patch_size, n_layers - 1,
                                                              min_size=min_size)
        if norm_name:
            if norm_name.lower() != "group":
                raise ValueError(f"Deprecating option 'norm_name={norm_name}', please use 'norm' instead.")
            norm = ("group", {"num_groups": num_groups})
        self.norm, self.upsample_mode, self.use_conv_final = norm, upsample_mode, use_conv_final
        self.convInit = get_dwconv_layer(spatial_dims, in_ch, mid_ch)
        self.down_layers = self._make_down_layers()
        self.up_layers, self.up_samples = self._make_up_layers()
        self.conv_final = self._make_final_conv(out_ch)
        if dropout_prob is not None:
            self.dropout = {2: nn.Dropout2d, 3: nn.Dropout3d}[spatial_dims](dropout_prob)

    def _make_down_layers(self):
        orders = ('d h w', 'd w h', 'w h d') if self.spatial_dims == 3 else ('h w', 'w h')
        down_layers = nn.ModuleList()
        for i, item in enumerate(self.blocks_down):
            ch = self.layer_in_channels[i]
            down = MaxPool(self.spatial_dims, kernel_size=self.scales[i], stride=self.scales[i]) \
                if np.prod(self.scales[i]) != 1 else nn.Identity()
            down_layers.append(nn.Sequential(down, *[self.block_cls(self.spatial_dims, ch, norm=self.norm, act=self.act,
                                                                    order=orders[i % len(orders)]) for _ in range(item)]))
        return down_layers

    def _make_up_layers(self):
        up_layers, up_samples = nn.ModuleList(), nn.ModuleList()
        sd = self.spatial_dims
        for i in range(len(self.blocks_up)):
            ch = self.layer_in_channels[i]
            up_layers.append(nn.Sequential(*[ResUpBlock(sd, ch, norm=self.norm, act=self.act)
                                             for _ in range(self.blocks_up[i])]))
            sc = self.scales[-(i + 1)]
            up_samples.append(nn.Sequential(
                Convolution(sd, ch, ch, strides=1, kernel_size=1, bias=False, conv_only=True),
                get_upsample_layer(sd, ch, upsample_mode=self.upsample_mode, scale_factor=sc)
                if np.prod(sc) != 1 else nn.Identity()))
        return up_layers, up_samples

    def _make_final_conv(self, out_channels: int):
        return nn.Sequential(get_norm_layer(self.norm, self.spatial_dims, self.init_filters), self.act_mod,
                             get_dwconv_layer(self.spatial_dims, self.init_filters, out_channels, kernel_size=1, bias=True))

    def encode(self, x):
        x = self.convInit(x)
        if self.dropout_prob is not None:
            x = self.dropout(x)
        down_x = []
        for down in self.down_layers:
            x = down(x)
            down_x.append(x)
        return x, down_x

    def decode(self, x, down_x):
        for i, (up_sample, upl) in enumerate(zip(self.up_samples, self.up_layers)):
            x = upl(up_sample(x) + down_x[i + 1])
        return self.conv_final(x) if self.use_conv_final else x

    def forward(self, x):
        last_add = self.rebnconvin(x) if (self.lm_variant and self.add_last) else None
        x, down_x = self.encode(x)
        down_x.reverse()
        x = self.decode(x, down_x)
        return x if last_add is None else x + last_add


class _LightX2(_UnetrStageX2):
    """outer wiring of both LightMamba2Net variants from per-stage channel tables; forward = _UnetrStageX2.forward (the
    reference's two forward methods :905-984, :1176-1255 are that same sequence)"""

    def _build_light(self, spatial_dims, deep_supervision, input_patch_size, out_ch, enc, dec, side_in, side_kernel,
                     concat_identity_ok):
        sd = self.spatial_dims = spatial_dims
        self.input_patch_size, self.deep_supervision = input_patch_size, deep_supervision
        self.scales = scales = get_scales(sd, input_patch_size, n_layers=5, patch_size=None, min_size=8)

        def ips(k):
            return input_patch_size if k == 0 else get_scale_value(sd, input_patch_size, scales[:k])

        n_layers = (7, 6, 5, 4, 4, 4)
        for i in range(6):
            setattr(self, f"stage{i + 1}", LightMUNet(spatial_dims=sd, **enc[i], n_layers=n_layers[i],
                                                      input_patch_size=ips(i), add_last=i < 5))
            if i < 5:
                setattr(self, f"patch_merging{i + 1}", PatchMerging2D(sd, enc[i]["out_ch"], scale=scales[i],
                                                                      output_features=enc[i + 1]["in_ch"]))
        e6 = enc[5]["out_ch"]
        self.patch_expand5d = PatchExpand(sd, dim=e6, scale=scales[-1], norm_layer=LayerNorm, output_dim=e6)
        for j, lvl in enumerate((5, 4, 3, 2, 1)):
            setattr(self, f"stage{lvl}d", LightMUNet(spatial_dims=sd, **dec[j], n_layers=n_layers[lvl - 1],
                                                     input_patch_size=ips(lvl - 1), add_last=True))
            if lvl > 1:
                half = dec[j]["out_ch"] // 2
                setattr(self, f"patch_expand{lvl - 1}d", PatchExpand(sd, dim=dec[j]["out_ch"], scale=scales[lvl - 2],
                                                                    norm_layer=LayerNorm, output_dim=half))
                nxt = dec[j + 1]["in_ch"]
                setattr(self, f"concat_back_dim{lvl - 1}d",
                        nn.Identity() if (concat_identity_ok and 2 * half == nxt) else TokenLinear(2 * half, nxt))
        for i, c in enumerate(side_in):
            setattr(self, f"side{i + 1}", Convolution(sd, c, out_ch, kernel_size=side_kernel,
                                                      padding=(side_kernel - 1) // 2, conv_only=True))
        self.outconv = Convolution(sd, 6 * out_ch, out_ch, kernel_size=1, conv_only=True)


class LightMamba2Net(_LightX2):
    def __init__(self, spatial_dims: int, in_ch: int, out_ch: int, deep_supervision: bool, input_patch_size):
        super().__init__()
        enc = [dict(in_ch=in_ch, mid_ch=16, out_ch=32), dict(in_ch=64, mid_ch=32, out_ch=64),
               dict(in_ch=128, mid_ch=64, out_ch=128), dict(in_ch=256, mid_ch=128, out_ch=256),
               dict(in_ch=512, mid_ch=256, out_ch=512), dict(in_ch=512, mid_ch=256, out_ch=512)]
        dec = [dict(in_ch=1024, mid_ch=256, out_ch=512), dict(in_ch=256, mid_ch=128, out_ch=256),
               dict(in_ch=128, mid_ch=64, out_ch=128), dict(in_ch=64, mid_ch=32, out_ch=64),
               dict(in_ch=32, mid_ch=16, out_ch=32)]
        self._build_light(spatial_dims, deep_supervision, input_patch_size, out_ch, enc, dec,
                          side_in=(32, 64, 128, 256, 512, 512), side_kernel=1, concat_identity_ok=False)


class LightMamba2NetP(_LightX2):
    def __init__(self, spatial_dims: int, in_ch: int, out_ch: int, deep_supervision: bool, input_patch_size):
        super().__init__()
        enc = [dict(in_ch=in_ch, mid_ch=32, out_ch=64)] + [dict(in_ch=64, mid_ch=32, out_ch=64) for _ in range(5)]
        dec = [dict(in_ch=128, mid_ch=32, out_ch=128) for _ in range(5)]
        self._build_light(spatial_dims, deep_supervision, input_patch_size, out_ch, enc, dec,
                          side_in=(128, 128, 128, 128, 128, 64), side_kernel=3, concat_identity_ok=True)


def get_light_mamba2net_from_plans(plans_manager, dataset_json: dict, configuration_manager, num_input_channels: int,
                                   deep_supervision: bool = True, use_pretrain: bool = True, small_model=False):
    cls = LightMamba2NetP if small_model else LightMamba2Net
    model = cls(spatial_dims=len(configuration_manager.patch_size), input_patch_size=configuration_manager.patch_size,
                in_ch=num_input_channels, out_ch=_heads(plans_manager, dataset_json), deep_supervision=deep_supervision)
    model.apply(InitWeights_He(1e-2))
    return model
