"""LM2Net / LM2NetP ("LM^2-Net": the 1-D-Mamba member of the LightMUNet family) - reference:
/root/reference/nnunetv2/nets/lm2net.py
  MambaLayer :64-92 (mamba_ssm.Mamba = the vendored nets/seg_mamba/mamba_simple.py:37-357), ResMambaBlock :107-176,
  ResUpBlock :179-220, LightMUNet :223-402, GSC :417-460, REBNCONV :463-476 (depthwise-separable), RSU4F :660-692,
  LM2Net :794-994, LM2NetP :1100-1309, get_lm2net_from_plans :1312-1343; trainers nnUNetTrainerLM2Net[P]
  (training/nnUNetTrainer/nnUNetTrainerLM2Net.py).

The file shares every helper with light_mamba2net.py (get_scales, PatchMerging2D, PatchExpand, GSC, ResMambaBlock are
character-identical there); what differs is stated in the subclasses below and in the outer wiring:
  * the mixer of MambaLayer is the 1-D Mamba block (nnuzoo_amd/nets/mamba_simple.py on the HIP causal-conv1d / scan / gate
    kernels) instead of Mamba2;
  * LightMUNet honours `add_last` (a depthwise-separable conv of the stage input, registered FIRST, added to the stage
    output) and has one ResMambaBlock per encoder level;
  * the two deepest encoder stages and the deepest decoder stage are RSU4F blocks (depthwise-separable REBNCONV, 2-D) with a
    ceil-mode max pool between them and a bilinear up-sampling back - the net is 2-D only (the reference permutes with
    fixed 4-D index tuples);
  * LM2NetP concatenates skip and expanded features channel-first and feeds 128 channels into every decoder stage; its
    stage2d / stage1d take `get_scale_value(..., scales[:2])` as their inner patch size (reference quirk :1186, :1201, kept:
    it only shapes the inner max-pool scales).
`model.apply(init_last_bn_before_add_to_0)` of the factory (:1337) touches residual BatchNorm blocks of
dynamic_network_architectures only - none exist here."""
from __future__ import annotations

import torch
from torch import nn

from ..layer_norm import LayerNorm
from ..token_linear import TokenLinear
from ..utilities.network_initialization import InitWeights_He
from . import light_mamba2net as _lm
from .common2d import Convolution, _upsample_like
from .mamba_nd2net import PatchExpand, PatchMerging2D
from .ssnd2net import _heads, get_scale_value, get_scales
from .swt2net import RSU4F


class MambaLayer(_lm.MambaLayer):
    mixer = "mamba"


class ResMambaBlock(_lm.ResMambaBlock):
    layer_cls = MambaLayer


class LightMUNet(_lm.LightMUNet):
    block_cls = ResMambaBlock
    lm_variant = True


class _LM2Base(nn.Module):
    def _encoder(self, sd, in_ch, input_patch_size, enc):
        """stages 1-4 (LightMUNet + PatchMerging2D), stages 5 / 6 RSU4F around a ceil-mode max pool"""
        self.scales = scales = get_scales(sd, input_patch_size, n_layers=5, patch_size=None, min_size=8)
        n_layers = (7, 6, 5, 4)
        for i in range(4):
            ips = input_patch_size if i == 0 else get_scale_value(sd, input_patch_size, scales[:i])
            setattr(self, f"stage{i + 1}", LightMUNet(spatial_dims=sd, **enc[i], n_layers=n_layers[i],
                                                      input_patch_size=ips, add_last=True))
            setattr(self, f"patch_merging{i + 1}", PatchMerging2D(sd, enc[i]["out_ch"], scale=scales[i],
                                                                  output_features=enc[i + 1]["in_ch"] if i < 3 else
                                                                  enc[4]))
        return scales

    def _encode(self, x):
        hx1 = self.stage1(x)
        hx2 = self.stage2(self.patch_merging1(hx1, permute_=True))
        hx3 = self.stage3(self.patch_merging2(hx2, permute_=True))
        hx4 = self.stage4(self.patch_merging3(hx3, permute_=True))
        hx5 = self.stage5(self.patch_merging4(hx4, permute_=True))
        hx6 = self.stage6(self.pool56(hx5))
        hx5d = self.stage5d(torch.cat((_upsample_like(hx6, hx5.shape[2:]), hx5), 1))
        return hx1, hx2, hx3, hx4, hx5d, hx6

    def _heads_out(self, dec, hx6):
        sides = [self.side1(dec[1]), self.side2(dec[2]), self.side3(dec[3]), self.side4(dec[4]), self.side5(dec[5]),
                 self.side6(hx6)]
        d0 = self.outconv(torch.cat([sides[0]] + [_upsample_like(s, sides[0].shape[2:]) for s in sides[1:]], 1))
        return (d0, *sides) if self.deep_supervision else d0

    def _encoder_groups(self):
        return [getattr(self, f"stage{i}") for i in range(1, 7)] + [getattr(self, f"patch_merging{i}") for i in range(1, 5)] \
            + [self.pool56]

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


class LM2Net(_LM2Base):
    def __init__(self, spatial_dims: int, in_ch: int, out_ch: int, deep_supervision: bool, input_patch_size):
        super().__init__()
        if spatial_dims != 2:
            raise NotImplementedError("LM2Net is a 2-D network (RSU4F stages, 4-D permutes: lm2net.py:832-838, :941-958)")
        sd = spatial_dims
        self.deep_supervision, self.input_patch_size = deep_supervision, input_patch_size
        enc = [dict(in_ch=in_ch, mid_ch=32, out_ch=32), dict(in_ch=64, mid_ch=32, out_ch=64),
               dict(in_ch=128, mid_ch=64, out_ch=128), dict(in_ch=256, mid_ch=128, out_ch=256), 512]
        scales = self._encoder(sd, in_ch, input_patch_size, enc)
        self.stage5 = RSU4F(512, 256, 512)
        self.pool56 = nn.MaxPool2d(2, stride=2, ceil_mode=True)
        self.stage6 = RSU4F(512, 256, 512)
        self.stage5d = RSU4F(1024, 256, 512)
        dec = {4: dict(in_ch=256, mid_ch=128, out_ch=256, n_layers=4), 3: dict(in_ch=128, mid_ch=64, out_ch=128, n_layers=5),
               2: dict(in_ch=64, mid_ch=32, out_ch=64, n_layers=6), 1: dict(in_ch=32, mid_ch=16, out_ch=32, n_layers=7)}
        prev = 512
        for lvl in (4, 3, 2, 1):
            c = dec[lvl]["out_ch"]
            setattr(self, f"patch_expand{lvl}d", PatchExpand(sd, dim=prev, scale=scales[lvl - 6], norm_layer=LayerNorm,
                                                             output_dim=c))
            setattr(self, f"concat_back_dim{lvl}d", TokenLinear(2 * c, c))
            ips = input_patch_size if lvl == 1 else get_scale_value(sd, input_patch_size, scales[:lvl - 1])
            setattr(self, f"stage{lvl}d", LightMUNet(spatial_dims=sd, **dec[lvl], input_patch_size=ips, add_last=True))
            prev = c
        for i, c in enumerate((32, 64, 128, 256, 512, 512)):
            setattr(self, f"side{i + 1}", Convolution(sd, c, out_ch, kernel_size=1, padding=0, conv_only=True))
        self.outconv = Convolution(sd, 6 * out_ch, out_ch, kernel_size=1, conv_only=True)

    def forward(self, x):
        hx1, hx2, hx3, hx4, hx5d, hx6 = self._encode(x)
        d, dec = hx5d, {5: hx5d}
        for lvl, skip in ((4, hx4), (3, hx3), (2, hx2), (1, hx1)):
            up = getattr(self, f"patch_expand{lvl}d")(d)                                               # channel last
            up = getattr(self, f"concat_back_dim{lvl}d")(torch.cat((up, skip.permute(0, 2, 3, 1)), -1)).permute(0, 3, 1, 2)
            d = getattr(self, f"stage{lvl}d")(up)
            dec[lvl] = d
        return self._heads_out(dec, hx6)


class LM2NetP(_LM2Base):
    def __init__(self, spatial_dims: int, in_ch: int, out_ch: int, deep_supervision: bool, input_patch_size):
        super().__init__()
        if spatial_dims != 2:
            raise NotImplementedError("LM2NetP is a 2-D network (RSU4F stages, 4-D permutes: lm2net.py:1138-1146, :1247-1262)")
        sd = spatial_dims
        self.deep_supervision, self.input_patch_size = deep_supervision, input_patch_size
        enc = [dict(in_ch=in_ch, mid_ch=32, out_ch=64)] + [dict(in_ch=64, mid_ch=32, out_ch=64) for _ in range(3)] + [64]
        scales = self._encoder(sd, in_ch, input_patch_size, enc)
        self.stage5 = RSU4F(64, 32, 64)
        self.pool56 = nn.MaxPool2d(2, stride=2, ceil_mode=True)
        self.stage6 = RSU4F(64, 32, 64)
        self.stage5d = RSU4F(128, 64, 128)
        n_layers = {4: 4, 3: 5, 2: 6, 1: 7}
        # inner patch sizes as the reference writes them: stage2d and stage1d both say scales[:2] (:1186, :1201)
        ips = {4: scales[:3], 3: scales[:2], 2: scales[:2], 1: scales[:2]}
        for lvl in (4, 3, 2, 1):
            setattr(self, f"patch_expand{lvl}d", PatchExpand(sd, dim=128, scale=scales[lvl - 6], norm_layer=LayerNorm,
                                                             output_dim=64))
            setattr(self, f"stage{lvl}d", LightMUNet(spatial_dims=sd, in_ch=128, mid_ch=32, out_ch=128,
                                                     n_layers=n_layers[lvl], add_last=True,
                                                     input_patch_size=get_scale_value(sd, input_patch_size, ips[lvl])))
        for i, c in enumerate((128, 128, 128, 128, 128, 64)):
            setattr(self, f"side{i + 1}", Convolution(sd, c, out_ch, kernel_size=1, padding=0, conv_only=True))
        self.outconv = Convolution(sd, 6 * out_ch, out_ch, kernel_size=1, conv_only=True)

    def forward(self, x):
        hx1, hx2, hx3, hx4, hx5d, hx6 = self._encode(x)
        d, dec = hx5d, {5: hx5d}
        for lvl, skip in ((4, hx4), (3, hx3), (2, hx2), (1, hx1)):
            up = getattr(self, f"patch_expand{lvl}d")(d)                                               # channel last
            d = getattr(self, f"stage{lvl}d")(torch.cat([up.permute(0, 3, 1, 2), skip], 1))
            dec[lvl] = d
        return self._heads_out(dec, hx6)


def get_lm2net_from_plans(plans_manager, dataset_json: dict, configuration_manager, num_input_channels: int,
                          deep_supervision: bool = True, use_pretrain: bool = True, small: bool = False):
    cls = LM2NetP if small else LM2Net
    model = cls(spatial_dims=len(configuration_manager.patch_size), input_patch_size=configuration_manager.patch_size,
                in_ch=num_input_channels, out_ch=_heads(plans_manager, dataset_json), deep_supervision=deep_supervision)
    model.apply(InitWeights_He(1e-2))
    return model
