"""The monai building blocks the reference's UNETR-style zoo nets import (`UnetrBasicBlock`, `UnetrPrUpBlock`,
`UnetrUpBlock`, `UnetOutBlock`: nets/mamba_nd2net.py:29-30, unetr2net.py, light_mamba2net.py), restated.

PARITY UNPINNED: monai is a third-party dependency absent from /root/reference and from this image, so
neither activations nor state_dict keys of these blocks can be checked against it here.  Structure, registration order
and parameter names follow the published monai 1.3 sources (monai/networks/blocks/unetr_block.py, dynunet_block.py):
  get_conv_layer(...)      -> `Convolution`: an nn.Sequential whose conv child is named `conv` (bias False unless asked)
  UnetResBlock             conv1, conv2, lrelu, norm1, norm2 [, conv3, norm3 when channels or stride change]
  UnetBasicBlock           conv1, conv2, lrelu, norm1, norm2
  UnetrBasicBlock.layer / UnetrUpBlock.{transp_conv, conv_block} / UnetrPrUpBlock.{transp_conv_init, blocks}
  UnetOutBlock.conv        1x1 conv with bias
norm_name "instance" is InstanceNorm without affine parameters (monai's default), activation LeakyReLU(0.01).
The arithmetic is stock torch convolutions / instance norms on the device (library kernels): these blocks are glue
around the state-space / attention cores, not the hot kernels of the path.
"""
from __future__ import annotations

from typing import Sequence, Union

import numpy as np
import torch
from torch import nn


def _tup(v, nd):
    return tuple(int(i) for i in (v if isinstance(v, (tuple, list)) else (v,) * nd))


def get_padding(kernel_size, stride, nd):
    k, s = np.array(_tup(kernel_size, nd)), np.array(_tup(stride, nd))
    p = (k - s + 1) / 2
    if np.min(p) < 0:
        raise AssertionError("padding value should not be negative, please change the kernel size and/or stride.")
    return tuple(int(i) for i in p)


def get_output_padding(kernel_size, stride, padding, nd):
    k, s, p = np.array(_tup(kernel_size, nd)), np.array(_tup(stride, nd)), np.array(_tup(padding, nd))
    o = 2 * p + s - k
    if np.min(o) < 0:
        raise AssertionError("out_padding value should not be negative, please change the kernel size and/or stride.")
    return tuple(int(i) for i in o)


class Convolution(nn.Sequential):
    """conv-only form of monai's Convolution block (no act / norm / dropout were requested by any caller here)"""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, bias=False, is_transposed=False):
        super().__init__()
        nd = spatial_dims
        pad = get_padding(kernel_size, stride, nd)
        if is_transposed:
            op = get_output_padding(kernel_size, stride, pad, nd)
            cls = {2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}[nd]
            conv = cls(in_channels, out_channels, _tup(kernel_size, nd), _tup(stride, nd), pad, op, bias=bias)
        else:
            cls = {2: nn.Conv2d, 3: nn.Conv3d}[nd]
            conv = cls(in_channels, out_channels, _tup(kernel_size, nd), _tup(stride, nd), pad, bias=bias)
        self.add_module("conv", conv)


def get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=3, stride=1, bias=False, is_transposed=False,
                   **_ignored):
    return Convolution(spatial_dims, in_channels, out_channels, kernel_size, stride, bias=bias,
                       is_transposed=is_transposed)


def get_norm_layer(name, spatial_dims, channels):
    kind, kw = (name, {}) if isinstance(name, str) else (name[0], dict(name[1]))
    kind = kind.lower()
    if kind == "instance":
        return {2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}[spatial_dims](channels, **kw)
    if kind == "batch":
        return {2: nn.BatchNorm2d, 3: nn.BatchNorm3d}[spatial_dims](channels, **kw)
    if kind == "group":
        return nn.GroupNorm(num_channels=channels, **kw)
    raise NotImplementedError(f"norm {name!r}")


class UnetResBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name):
        super().__init__()
        self.conv1 = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size, stride)
        self.conv2 = get_conv_layer(spatial_dims, out_channels, out_channels, kernel_size, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=0.01, inplace=True)
        self.norm1 = get_norm_layer(norm_name, spatial_dims, out_channels)
        self.norm2 = get_norm_layer(norm_name, spatial_dims, out_channels)
        self.downsample = in_channels != out_channels
        if not np.all(np.atleast_1d(stride) == 1):
            self.downsample = True
        if self.downsample:
            self.conv3 = get_conv_layer(spatial_dims, in_channels, out_channels, 1, stride)
            self.norm3 = get_norm_layer(norm_name, spatial_dims, out_channels)

    def forward(self, inp):
        residual = inp
        out = self.lrelu(self.norm1(self.conv1(inp)))
        out = self.norm2(self.conv2(out))
        if hasattr(self, "conv3"):
            residual = self.norm3(self.conv3(residual))
        out = out + residual
        return self.lrelu(out)


class UnetBasicBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name):
        super().__init__()
        self.conv1 = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size, stride)
        self.conv2 = get_conv_layer(spatial_dims, out_channels, out_channels, kernel_size, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=0.01, inplace=True)
        self.norm1 = get_norm_layer(norm_name, spatial_dims, out_channels)
        self.norm2 = get_norm_layer(norm_name, spatial_dims, out_channels)

    def forward(self, inp):
        out = self.lrelu(self.norm1(self.conv1(inp)))
        return self.lrelu(self.norm2(self.conv2(out)))


class UnetOutBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, dropout=None):
        super().__init__()
        self.conv = get_conv_layer(spatial_dims, in_channels, out_channels, 1, 1, bias=True)

    def forward(self, inp):
        return self.conv(inp)


class UnetrBasicBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name, res_block=False):
        super().__init__()
        blk = UnetResBlock if res_block else UnetBasicBlock
        self.layer = blk(spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name)

    def forward(self, inp):
        return self.layer(inp)


class UnetrUpBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, upsample_kernel_size, norm_name,
                 res_block=False):
        super().__init__()
        self.transp_conv = get_conv_layer(spatial_dims, in_channels, out_channels, upsample_kernel_size,
                                          upsample_kernel_size, is_transposed=True)
        blk = UnetResBlock if res_block else UnetBasicBlock
        self.conv_block = blk(spatial_dims, out_channels + out_channels, out_channels, kernel_size, 1, norm_name)

    def forward(self, inp, skip):
        out = self.transp_conv(inp)
        return self.conv_block(torch.cat((out, skip), dim=1))


class UnetrPrUpBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, num_layer, kernel_size, stride, upsample_kernel_size,
                 norm_name, conv_block=False, res_block=False):
        super().__init__()
        up = upsample_kernel_size
        self.transp_conv_init = get_conv_layer(spatial_dims, in_channels, out_channels, up, up, is_transposed=True)
        if conv_block:
            blk = UnetResBlock if res_block else UnetBasicBlock
            self.blocks = nn.ModuleList([nn.Sequential(
                get_conv_layer(spatial_dims, out_channels, out_channels, up, up, is_transposed=True),
                blk(spatial_dims, out_channels, out_channels, kernel_size, stride, norm_name)) for _ in range(num_layer)])
        else:
            self.blocks = nn.ModuleList([get_conv_layer(spatial_dims, out_channels, out_channels, up, up,
                                                        is_transposed=True) for _ in range(num_layer)])

    def forward(self, x):
        x = self.transp_conv_init(x)
        for blk in self.blocks:
            x = blk(x)
        return x
