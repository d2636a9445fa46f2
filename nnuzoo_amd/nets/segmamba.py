"""SegMamba - the zoo's 3-D Mamba segmentation net (also runs 2-D) - for MI355X.

Same classes, constructor arguments, registration order and state_dict keys as the reference's
/root/reference/nnunetv2/nets/seg_mamba/segmamba.py:27-372 (`InstanceNorm`, `LayerNorm`, `MambaLayer`, `MlpChannel`, `GSC`,
`MambaEncoder`, `SegMamba`, `get_seg_mamba_from_plans`); trainer plugin `nnUNetTrainerSegMamba`
(training/nnUNetTrainer/nnUNetTrainerSegMamba.py) in nnuzoo_amd/training/zoo_trainers.py.

What runs where:
  * `MambaLayer` (:65-90): LayerNorm -> Mamba (bimamba "v3" in 3-D: forward + backward + slice direction, "v2" in 2-D) ->
    residual.  The block is nnuzoo_amd.nets.mamba_simple.Mamba on the HIP operators of nnuzoo_amd.mamba_block (causal conv1d +
    SiLU, selective scan with z gate: csrc/mamba_block.hip, csrc/selective_scan.hip), the norm is nnuzoo_amd.layer_norm.
  * `GSC`, `MlpChannel`, the stem / down-sampling convolutions: stock torch convolutions + instance norms (library kernels on
    the device) - small-channel glue around the state-space core, like the reference.
  * The UNETR-style encoder / decoder blocks come from monai in the reference (`UnetrBasicBlock`, `UnetrUpBlock`, `UnetOutBlock`,
    segmamba.py:20-21); monai is absent here: nnuzoo_amd/nets/monai_blocks.py restates them (PARITY UNPINNED for those blocks,
    see that file's header).  Everything this file defines itself - the whole `MambaEncoder` - is pinned against the
    reference's own module (tests/golden/segmamba_encoder_*.npz, tools/make_golden_segmamba.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..layer_norm import LayerNorm as _HipLayerNorm
from ..utilities.network_initialization import InitWeights_He
from .mamba_simple import Mamba
from .monai_blocks import Convolution as _MonaiConv, UnetOutBlock, UnetrBasicBlock, UnetrUpBlock


def _conv(spatial_dims, cin, cout, kernel_size, strides=1, padding=None):
    """monai `Convolution(..., conv_only=True)`: an nn.Sequential with one child named `conv` (bias True - monai's default for
    the block, unlike get_conv_layer); explicit padding as the reference passes it."""
    seq = _MonaiConv(spatial_dims, cin, cout, kernel_size, strides, bias=True)
    if padding is not None:
        seq.conv.padding = (padding,) * spatial_dims
    return seq


class InstanceNorm(nn.Module):
    """segmamba.py:27-37 (no affine parameters: torch's default)"""

    def __init__(self, spatial_dims: int, in_channels: int):
        super().__init__()
        self.layer = {2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}[spatial_dims](in_channels)

    def forward(self, input):
        return self.layer(input)


class LayerNorm(_HipLayerNorm):
    """The `LayerNorm(normalized_shape, eps, data_format)` name of segmamba.py:40-62 on the shared HIP LayerNorm
    (nnuzoo_amd/layer_norm.py): channels_last normalises the trailing axis as it stands; channels_first moves the channel axis
    behind the spatial ones (the kernel packs the tokens), normalises, and moves it back.  SegMamba itself never
    instantiates it; the name is part of the module's surface."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        super().__init__(normalized_shape, eps=eps)
        self.data_format = data_format

    def forward(self, x):
        if self.data_format == "channels_last":
            return super().forward(x)
        return super().forward(x.movedim(1, -1)).movedim(-1, 1)


class MambaLayer(nn.Module):
    """segmamba.py:65-90: tokens = flattened voxels, LayerNorm -> Mamba -> + skip"""

    def __init__(self, spatial_dims: int, dim, d_state=16, d_conv=4, expand=2, num_slices=None):
        super().__init__()
        self.dim = dim
        self.norm = _HipLayerNorm(dim)
        self.mamba = Mamba(d_model=dim, d_state=d_state, d_conv=d_conv, expand=expand,
                           bimamba_type="v3" if spatial_dims == 3 else "v2", nslices=num_slices)

    def forward(self, x):
        B, C = x.shape[:2]
        assert C == self.dim
        img_dims = x.shape[2:]
        n_tokens = img_dims.numel()
        x_flat = x.reshape(B, C, n_tokens).transpose(-1, -2)
        # the scan kernels and the LayerNorm kernel are fp32 (the reference's Mamba block computes in fp32 as well: its
        # selective_scan_fn casts, mamba_simple.py); under autocast only the projections' GEMMs run in fp16
        x_mamba = self.mamba(self.norm(x_flat.float().contiguous()))
        out = x_mamba.transpose(-1, -2).reshape(B, C, *img_dims)
        return out.to(x.dtype) + x


class MlpChannel(nn.Module):
    """segmamba.py:93-104: 1x1 conv -> GELU -> 1x1 conv"""

    def __init__(self, spatial_dims, hidden_size, mlp_dim):
        super().__init__()
        self.fc1 = _conv(spatial_dims, hidden_size, mlp_dim, 1)
        self.act = nn.GELU()
        self.fc2 = _conv(spatial_dims, mlp_dim, hidden_size, 1)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class GSC(nn.Module):
    """segmamba.py:107-155: gated spatial convolution (two 3x3 + one 1x1 branch, 1x1 merge, each conv -> InstanceNorm -> ReLU)"""

    def __init__(self, spatial_dims: int, in_channles) -> None:
        super().__init__()
        c = in_channles
        self.proj = _conv(spatial_dims, c, c, 3, 1, padding=1)
        self.norm = InstanceNorm(spatial_dims, c)
        self.nonliner = nn.ReLU()
        self.proj2 = _conv(spatial_dims, c, c, 3, 1, padding=1)
        self.norm2 = InstanceNorm(spatial_dims, c)
        self.nonliner2 = nn.ReLU()
        self.proj3 = _conv(spatial_dims, c, c, 1, 1, padding=0)
        self.norm3 = InstanceNorm(spatial_dims, c)
        self.nonliner3 = nn.ReLU()
        self.proj4 = _conv(spatial_dims, c, c, 1, 1, padding=0)
        self.norm4 = InstanceNorm(spatial_dims, c)
        self.nonliner4 = nn.ReLU()

    def forward(self, x):
        x1 = self.nonliner(self.norm(self.proj(x)))
        x1 = self.nonliner2(self.norm2(self.proj2(x1)))
        x2 = self.nonliner3(self.norm3(self.proj3(x)))
        y = self.nonliner4(self.norm4(self.proj4(x1 + x2)))
        return y + x


class MambaEncoder(nn.Module):
    """segmamba.py:158-222: k7 s2 stem, three InstanceNorm + k2 s2 down-samplings; per level GSC -> `depths[i]` MambaLayers;
    outputs = InstanceNorm -> MlpChannel of every level"""

    def __init__(self, spatial_dims: int, in_chans=1, depths=[2, 2, 2, 2], dims=[48, 96, 192, 384], drop_path_rate=0.,
                 layer_scale_init_value=1e-6, out_indices=[0, 1, 2, 3]):
        super().__init__()
        self.downsample_layers = nn.ModuleList()
        self.downsample_layers.append(nn.Sequential(_conv(spatial_dims, in_chans, dims[0], 7, 2, padding=3)))
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(InstanceNorm(spatial_dims, dims[i]),
                                                        _conv(spatial_dims, dims[i], dims[i + 1], 2, 2, padding=0)))
        self.stages = nn.ModuleList()
        self.gscs = nn.ModuleList()
        num_slices_list = [64, 32, 16, 8]
        for i in range(4):
            gsc = GSC(spatial_dims, dims[i])
            stage = nn.Sequential(*[MambaLayer(spatial_dims, dim=dims[i], num_slices=num_slices_list[i])
                                    for _ in range(depths[i])])
            self.stages.append(stage)
            self.gscs.append(gsc)
        self.out_indices = out_indices
        self.mlps = nn.ModuleList()
        for i_layer in range(4):
            self.add_module(f'norm{i_layer}', InstanceNorm(spatial_dims, dims[i_layer]))
            self.mlps.append(MlpChannel(spatial_dims, dims[i_layer], 2 * dims[i_layer]))

    def forward(self, x):
        outs = []
        for i in range(4):
            x = self.downsample_layers[i](x)
            x = self.gscs[i](x)
            x = self.stages[i](x)
            if i in self.out_indices:
                outs.append(self.mlps[i](getattr(self, f'norm{i}')(x)))
        return tuple(outs)


class SegMamba(nn.Module):
    """segmamba.py:225-372"""

    def __init__(self, in_ch=1, out_ch=13, depths=[2, 2, 2, 2], feat_size=[48, 96, 192, 384], drop_path_rate=0,
                 layer_scale_init_value=1e-6, hidden_size: int = 768, norm_name="instance", res_block: bool = True,
                 spatial_dims=3) -> None:
        super().__init__()
        self.hidden_size, self.in_ch, self.out_ch, self.depths = hidden_size, in_ch, out_ch, depths
        self.drop_path_rate, self.feat_size, self.layer_scale_init_value = drop_path_rate, feat_size, layer_scale_init_value
        self.spatial_dims = spatial_dims
        sd, f = spatial_dims, feat_size
        self.vit = MambaEncoder(spatial_dims=sd, in_chans=in_ch, depths=depths, dims=feat_size, drop_path_rate=drop_path_rate,
                                layer_scale_init_value=layer_scale_init_value)

        def basic(cin, cout):
            return UnetrBasicBlock(spatial_dims=sd, in_channels=cin, out_channels=cout, kernel_size=3, stride=1,
                                   norm_name=norm_name, res_block=res_block)

        def up(cin, cout):
            return UnetrUpBlock(spatial_dims=sd, in_channels=cin, out_channels=cout, kernel_size=3, upsample_kernel_size=2,
                                norm_name=norm_name, res_block=res_block)

        self.encoder1 = basic(in_ch, f[0])
        self.encoder2 = basic(f[0], f[1])
        self.encoder3 = basic(f[1], f[2])
        self.encoder4 = basic(f[2], f[3])
        self.encoder5 = basic(f[3], hidden_size)
        self.decoder5 = up(hidden_size, f[3])
        self.decoder4 = up(f[3], f[2])
        self.decoder3 = up(f[2], f[1])
        self.decoder2 = up(f[1], f[0])
        self.decoder1 = basic(f[0], f[0])
        self.out = UnetOutBlock(spatial_dims=sd, in_channels=48, out_channels=out_ch)

    def forward(self, x_in):
        outs = self.vit(x_in)
        enc1 = self.encoder1(x_in)
        enc2 = self.encoder2(outs[0])
        enc3 = self.encoder3(outs[1])
        enc4 = self.encoder4(outs[2])
        enc_hidden = self.encoder5(outs[3])
        dec3 = self.decoder5(enc_hidden, enc4)
        dec2 = self.decoder4(dec3, enc3)
        dec1 = self.decoder3(dec2, enc2)
        dec0 = self.decoder2(dec1, enc1)
        return self.out(self.decoder1(dec0))


def get_seg_mamba_from_plans(plans_manager, dataset_json: dict, configuration_manager, num_input_channels: int,
                             deep_supervision: bool = True, use_pretrain: bool = True, small_mode: bool = False):
    """segmamba.py:375-411 (small_mode raises there as well; `InitWeights_He(1e-2)` then the no-op residual-BN init)"""
    if small_mode:
        raise NotImplementedError()
    from ..training.nnUNetTrainer import _num_segmentation_heads
    if plans_manager is not None and hasattr(plans_manager, "get_label_manager"):
        heads = plans_manager.get_label_manager(dataset_json).num_segmentation_heads
    else:
        heads = _num_segmentation_heads(dataset_json)
    model = SegMamba(spatial_dims=len(configuration_manager.patch_size), in_ch=num_input_channels, out_ch=heads)
    model.apply(InitWeights_He(1e-2))
    return model
