"""LightMUNet (the stand-alone "LightM-UNet") - reference: /root/reference/nnunetv2/nets/LightMUNet.py
  MambaLayer :49-77, get_mamba_layer :80-90, ResMambaBlock :93-161, ResUpBlock :164-205, LightMUNet :208-389,
  GSC :404-447, get_from_plans :450-487; trainer training/nnUNetTrainer/nnUNetTrainerLightMUNet.py.

A channel-doubling residual U-Net (2-D or 3-D): depthwise-separable input conv, four encoder levels of 1 / 2 / 2 / 4
ResMambaBlocks (GSC gate + two 1-D Mamba layers walking the voxels in the level's axis order), each level after the first
entered through a MambaLayer that doubles the channels followed by a 2 x max pool; decoder = 1x1 conv + (bi/tri)linear
up-sampling + skip sum + ResUpBlock per level; GroupNorm(8) + ReLU + depthwise-separable 1x1 head.  One output (the trainer
switches deep supervision off).

ResMambaBlock / ResUpBlock / MambaLayer / GSC are character-identical to the classes of lm2net.py (and therefore pinned by
the LM2Net fixtures as well); this file adds the outer class.  The 1-D mixer is nnuzoo_amd/nets/mamba_simple.py on the HIP
causal-conv1d / scan / gate kernels.  The masked-auto-encoder option (`mae`, mask_funcs.py) is not built: it raises."""
from __future__ import annotations

import torch
from torch import nn

from ..utilities.network_initialization import InitWeights_He
from .common2d import Convolution, get_dwconv_layer
from .light_mamba2net import _GROUP8, _RELU, ResUpBlock, get_act_layer, get_norm_layer, get_upsample_layer
from .lm2net import MambaLayer, ResMambaBlock


def get_mamba_layer(spatial_dims: int, in_channels: int, out_channels: int, stride: int = 1):
    layer = MambaLayer(input_dim=in_channels, output_dim=out_channels)
    if stride != 1:
        return nn.Sequential(layer, {2: nn.MaxPool2d, 3: nn.MaxPool3d}[spatial_dims](kernel_size=stride, stride=stride))
    return layer


class LightMUNet(nn.Module):
    def __init__(self, spatial_dims: int = 3, init_filters: int = 32, in_channels: int = 1, out_channels: int = 2,
                 dropout_prob=None, act=_RELU, norm=_GROUP8, norm_name: str = "", num_groups: int = 8,
                 use_conv_final: bool = True, blocks_down=(1, 2, 2, 4), blocks_up=(1, 1, 1),
                 upsample_mode="nontrainable", mae: bool = False, mask_ratio: float = 0.0):
        super().__init__()
        if spatial_dims not in (2, 3):
            raise ValueError("`spatial_dims` can only be 2 or 3.")
        if mae:
            raise NotImplementedError("LightMUNet(mae=True): the masked-auto-encoder pre-training path is not built")
        self.mae, self.spatial_dims = mae, spatial_dims
        self.init_filters, self.in_channels = init_filters, in_channels
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
        orders = ('d h w', 'd w h', 'w h d') if self.spatial_dims == 3 else ('h w', 'w h')
        down_layers = nn.ModuleList()
        for i, item in enumerate(self.blocks_down):
            ch = self.init_filters * 2 ** i
            down = get_mamba_layer(self.spatial_dims, ch // 2, ch, stride=2) if i > 0 else nn.Identity()
            down_layers.append(nn.Sequential(down, *[ResMambaBlock(self.spatial_dims, ch, norm=self.norm, act=self.act,
                                                                   order=orders[i % len(orders)]) for _ in range(item)]))
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
        x, down_x = self.encode(self.convInit(x))
        down_x.reverse()
        return self.decode(x, down_x)


def get_from_plans(spatial_dims: int, in_ch: int, out_ch: int, small_mode=False, mae: bool = False, mask_ratio: float = 0,
                   **kwargs):
    if small_mode:
        raise NotImplementedError()
    model = LightMUNet(spatial_dims=spatial_dims, in_channels=in_ch, out_channels=out_ch, mae=mae, mask_ratio=mask_ratio)
    model.apply(InitWeights_He(1e-2))
    return model
