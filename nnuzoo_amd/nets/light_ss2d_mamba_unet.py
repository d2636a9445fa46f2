"""LightSS2DMambaUNet - the LightM-UNet whose Mamba layers are SS2D blocks (2-D) - for MI355X.

Reference: /root/reference/nnunetv2/nets/LightSS2DMambaUNet.py: `MambaLayer` :281-312 (LayerNorm -> SS2D -> scaled skip ->
LayerNorm -> Linear on the token-major image), `get_mamba_layer` :315-324, `ResMambaBlock` :327-375 (GSC gate + two MambaLayers
named conv1 / conv2), `ResUpBlock` :378-419, `LightSS2DMambaUNet` :422-548, `get_mamband2net_from_plans` :551-583; trainer
training/nnUNetTrainer/nnUNetTrainerLightSS2DMambaUNet.py.

The file's `SS2D` (:77-263) is the class of nets/m2net.py (only argument spellings differ), so the mixer here IS
nnuzoo_amd.nets.m2net.SS2D: depthwise conv + SiLU (csrc/ss2d_dwconv.hip), four-direction selective scan
(csrc/selective_scan.hip / ss2d_scan_rl.hpp), LayerNorm + gate (csrc/layer_norm.hip), token-major Linear layers
(csrc/token_linear.hip).  GSC / ResUpBlock / the separable convolutions / the monai helper layers are the ones of
nnuzoo_amd/nets/light_mamba2net.py (same classes in the reference's file family).  MambaLayer.forward unpacks (B, C, H, W): the
network runs 2-D inputs only, like the reference's.  Pinned against the reference's own class: state_dict manifest + whole-net
forward / dx (tests/golden/net_LightSS2DMambaUNet_2d.npz, tools/make_golden_lm2net.py lightss2d); monai's get_conv_layer /
get_upsample_layer / get_norm_layer / get_act_layer are restated identically on both sides (unpinned)."""
from __future__ import annotations

import torch
from torch import nn

from ..layer_norm import LayerNorm
from ..token_linear import TokenLinear
from ..utilities.network_initialization import InitWeights_He
from .common2d import Convolution, get_dwconv_layer
from .light_mamba2net import _GROUP8, _RELU, GSC, ResUpBlock, get_act_layer, get_norm_layer, get_upsample_layer
from .m2net import SS2D


class MambaLayer(nn.Module):
    def __init__(self, input_dim, output_dim, d_state=16, d_conv=4, expand=2):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.input_norm = LayerNorm(input_dim)
        self.mamba = SS2D(d_model=input_dim, d_state=d_state)
        self.output_norm = LayerNorm(input_dim)
        self.proj = TokenLinear(input_dim, output_dim)
        self.skip_scale = nn.Parameter(torch.ones(1))

    def forward(self, x):
        if x.dtype == torch.float16:
            x = x.type(torch.float32)
        B, C, H, W = x.shape
        assert C == self.input_dim
        xt = x.permute(0, 2, 3, 1)
        xm = self.mamba(self.input_norm(xt)) + self.skip_scale * xt
        return self.proj(self.output_norm(xm)).permute(0, 3, 1, 2)


def get_mamba_layer(spatial_dims: int, in_channels: int, out_channels: int, stride: int = 1):
    layer = MambaLayer(input_dim=in_channels, output_dim=out_channels)
    if stride != 1:
        return nn.Sequential(layer, {2: nn.MaxPool2d, 3: nn.MaxPool3d}[spatial_dims](kernel_size=stride, stride=stride))
    return layer


class ResMambaBlock(nn.Module):
    def __init__(self, spatial_dims: int, in_channels: int, norm, kernel_size: int = 3, act=_RELU):
        super().__init__()
        if kernel_size % 2 != 1:
            raise AssertionError("kernel_size should be an odd number.")
        self.gsc = GSC(spatial_dims, in_channels)
        self.norm1 = get_norm_layer(norm, spatial_dims, in_channels)
        self.norm2 = get_norm_layer(norm, spatial_dims, in_channels)
        self.act = get_act_layer(act)
        self.conv1 = get_mamba_layer(spatial_dims, in_channels=in_channels, out_channels=in_channels)
        self.conv2 = get_mamba_layer(spatial_dims, in_channels=in_channels, out_channels=in_channels)

    def forward(self, x):
        x = self.gsc(x)
        identity = x
        x = self.conv1(self.act(self.norm1(x)))
        x = self.conv2(self.act(self.norm2(x)))
        return x + identity


class LightSS2DMambaUNet(nn.Module):
    def __init__(self, spatial_dims: int = 3, init_filters: int = 8, in_channels: int = 1, out_channels: int = 2,
                 dropout_prob=None, act=_RELU, norm=_GROUP8, norm_name: str = "", num_groups: int = 8,
                 use_conv_final: bool = True, blocks_down=(1, 2, 2, 4), blocks_up=(1, 1, 1), upsample_mode="nontrainable"):
        super().__init__()
        if spatial_dims not in (2, 3):
            raise ValueError("`spatial_dims` can only be 2 or 3.")
        self.spatial_dims, self.init_filters, self.in_channels = spatial_dims, init_filters, in_channels
        self.blocks_down, self.blocks_up = blocks_down, blocks_up
        self.dropout_prob, self.act = dropout_prob, act
        self.act_mod = get_act_layer(act)
        if norm_name:
            if norm_name.lower() != "group":
                raise ValueError(f"Deprecating option 'norm_name={norm_name}', please use 'norm' instead.")
            norm = ("group", {"num_groups": num_groups})
        self.norm, self.upsample_mode, self.use_conv_final = norm, upsample_mode, use_conv_final
        self.convInit = get_dwconv_layer(spatial_dims, in_channels, init_filters)
        self.down_layers = self._make_down_layers()
        self.up_layers, self.up_samples = self._make_up_layers()
        self.conv_final = self._make_final_conv(out_channels)
        if dropout_prob is not None:
            self.dropout = {2: nn.Dropout2d, 3: nn.Dropout3d}[spatial_dims](dropout_prob)

    def _make_down_layers(self):
        down_layers = nn.ModuleList()
        for i, item in enumerate(self.blocks_down):
            ch = self.init_filters * 2 ** i
            down = get_mamba_layer(self.spatial_dims, ch // 2, ch, stride=2) if i > 0 else nn.Identity()
            down_layers.append(nn.Sequential(down, *[ResMambaBlock(self.spatial_dims, ch, norm=self.norm, act=self.act)
                                                     for _ in range(item)]))
        return down_layers

    def _make_up_layers(self):
        up_layers, up_samples = nn.ModuleList(), nn.ModuleList()
        sd, n_up = self.spatial_dims, len(self.blocks_up)
        for i in range(n_up):
            ch = self.init_filters * 2 ** (n_up - i)
            up_layers.append(nn.Sequential(*[ResUpBlock(sd, ch // 2, norm=self.norm, act=self.act)
                                             for _ in range(self.blocks_up[i])]))
            up_samples.append(nn.Sequential(Convolution(sd, ch, ch // 2, strides=1, kernel_size=1, bias=False, conv_only=True),
                                            get_upsample_layer(sd, ch // 2, upsample_mode=self.upsample_mode)))
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
        for i, (up, upl) in enumerate(zip(self.up_samples, self.up_layers)):
            x = upl(up(x) + down_x[i + 1])
        return self.conv_final(x) if self.use_conv_final else x

    def forward(self, x):
        x, down_x = self.encode(x)
        down_x.reverse()
        return self.decode(x, down_x)


def get_mamband2net_from_plans(spatial_dims: int, in_ch: int, out_ch: int, small_mode=False, **kwargs):
    """the factory of the reference's file (its name is a left-over of the file it was copied from); `small_mode` raises there too"""
    if small_mode:
        raise NotImplementedError()
    model = LightSS2DMambaUNet(spatial_dims=spatial_dims, in_channels=in_ch, out_channels=out_ch)
    model.apply(InitWeights_He(1e-2))
    return model
