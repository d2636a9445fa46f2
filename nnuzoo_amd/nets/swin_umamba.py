"""Swin-UMamba and Swin-UMamba-D (2-D) for MI355X.

Same classes, constructor arguments, registration order and state_dict keys as the reference's
  /root/reference/nnunetv2/nets/SwinUMamba.py:367-695   `VSSMEncoder`, `SwinUMamba`, `load_pretrained_ckpt`, `get_swin_umamba_from_plans`
  /root/reference/nnunetv2/nets/SwinUMambaD.py:49-732   `PatchExpand`, `FinalPatchExpand_X4`, `UNetResDecoder`, `SwinUMambaD`,
                                                         `get_swin_umamba_d_from_plans`
trainer plugins `nnUNetTrainerSwinUMamba` / `nnUNetTrainerSwinUMambaD` in nnuzoo_amd/training/zoo_trainers.py.

Both files of the reference carry their own copies of `PatchEmbed2D`, `PatchMerging2D`, `SS2D`, `VSSBlock` and `VSSLayer`; they are
the classes of nets/m2net.py with other argument spellings (`ssm_ratio` for `expand`, the fixed 2 x 2 `PatchMerging2D(dim)`), so
the state-space core here IS the one of nnuzoo_amd/nets/m2net.py: four-direction selective scan (csrc/selective_scan.hip),
depthwise conv + SiLU, LayerNorm + gate (csrc/layer_norm.hip), token-major Linear layers on MFMA (csrc/token_linear.hip), and
the RNG-advancing no-op re-initialisation of `VSSLayer` replayed so that a seeded construction draws the reference's stream.

  * Swin-UMamba-D is defined entirely inside the reference's file (VSSM encoder, patch-expanding Mamba decoder, 1x1 heads): the
    whole network is pinned against the reference's own module (tests/golden/swin_umamba_d.npz, tools/make_golden_swin_umamba.py).
  * Swin-UMamba wraps the same encoder (patch size 2 behind a k7 s2 stem) in monai's UNETR blocks (`UnetrBasicBlock`,
    `UnetrUpBlock`, `UnetOutBlock`, SwinUMamba.py:17-18); monai is absent here: nnuzoo_amd/nets/monai_blocks.py restates them
    (PARITY UNPINNED for those blocks, see that file's header).
"""
from __future__ import annotations

import math
import re
from typing import List, Tuple, Union

import torch
import torch.nn as nn

from ..layer_norm import LayerNorm
from ..token_linear import TokenLinear
from ..utilities.network_initialization import InitWeights_He
from . import m2net as _m2
from .common2d import PatchExpand as _PatchExpand2
from .m2net import SS2D, VSSBlock, VSSLayer, PatchEmbed2D  # noqa: F401  (re-exported: the reference defines them in these files)
from .monai_blocks import UnetOutBlock, UnetrBasicBlock, UnetrUpBlock


class PatchMerging2D(_m2.PatchMerging2D):
    """SwinUMamba.py:47-87: 2 x 2 space-to-depth (channel blocks (0,0), (1,0), (0,1), (1,1)) -> LayerNorm(4 dim) -> Linear(2 dim)"""

    def __init__(self, dim, norm_layer=LayerNorm):
        super().__init__(input_dim=dim, scale=2, output_features=2 * dim, norm_layer=norm_layer)
        self.dim = dim


class VSSMEncoder(_m2.VSSMEncoder):
    """SwinUMamba.py:367-453 / SwinUMambaD.py:430-527: the VSSM encoder of nets/m2net.py without its U^2 options, stochastic depth
    0.2, and the INPUT as entry 0 of the returned list (the m2net form returns None there)."""

    def __init__(self, patch_size=4, in_chans=3, depths=[2, 2, 9, 2], dims=[96, 192, 384, 768], d_state=16, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.2, norm_layer=LayerNorm, patch_norm=True, use_checkpoint=False, **kwargs):
        super().__init__(patch_size=patch_size, in_chans=in_chans, depths=depths, dims=dims, d_state=d_state,
                         drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate,
                         norm_layer=norm_layer, patch_norm=patch_norm, use_checkpoint=use_checkpoint)
        self.num_features = self.dims[-1]
        self.ape = False

    def forward(self, x):
        feats = super().forward(x)
        feats[0] = x
        return feats


class PatchExpand(_PatchExpand2):
    """SwinUMambaD.py:49-71: NCHW in; Linear(dim -> 2 dim), 2 x 2 depth-to-space (dim / 2 channels), LayerNorm; token-major out"""

    def __init__(self, input_resolution, dim, dim_scale=2, norm_layer=LayerNorm):
        if dim_scale != 2:
            raise NotImplementedError("the reference's dim_scale != 2 branch (nn.Identity expand) cannot run: its view needs 4 | C")
        super().__init__(dim, 2, None, norm_layer)


class FinalPatchExpand_X4(nn.Module):
    """SwinUMambaD.py:74-108: NCHW in; Linear(dim -> 16 dim), 4 x 4 depth-to-space back to dim channels, LayerNorm; token-major out"""

    def __init__(self, input_resolution, dim, dim_scale=4, norm_layer=LayerNorm):
        super().__init__()
        self.dim, self.dim_scale = dim, dim_scale
        self.expand = TokenLinear(dim, 16 * dim, bias=False)
        self.output_dim = dim
        self.norm = norm_layer(self.output_dim)

    def forward(self, x):
        x = self.expand(x.permute(0, 2, 3, 1))
        B, H, W, C = x.shape
        s = self.dim_scale
        c = C // (s * s)
        x = x.view(B, H, W, s, s, c).permute(0, 1, 3, 2, 4, 5).reshape(B, H * s, W * s, c)
        return self.norm(x)


class UNetResDecoder(nn.Module):
    """SwinUMambaD.py:530-638.  Per stage, bottom up: PatchExpand of the map below, concatenation with the encoder skip on the
    channel axis, Linear(2 c -> c), two VSSBlocks, a 1x1 head; last: FinalPatchExpand_X4 back to the input resolution + head.
    Modules are CONSTRUCTED stage by stage (expand, VSSLayer, head, Linear - the order of the random draws) and REGISTERED as
    `stages`, `expand_layers`, `seg_layers`, `concat_back_dim` (the order of the state_dict), as there."""

    def __init__(self, num_classes: int, deep_supervision, features_per_stage: Union[Tuple[int, ...], List[int]] = [96, 192, 384, 768],
                 drop_path_rate: float = 0.2, d_state: int = 16):
        super().__init__()
        ch = features_per_stage
        self.deep_supervision = deep_supervision
        self.num_classes = num_classes
        n = len(ch)
        dpr = [x.item() for x in torch.linspace(drop_path_rate, 0, (n - 1) * 2)]
        depths = [2, 2, 2, 2]
        stages, expand_layers, seg_layers, concat_back_dim = [], [], [], []
        for s in range(1, n):
            below, skip = ch[-s], ch[-(s + 1)]
            expand_layers.append(PatchExpand(input_resolution=None, dim=below, dim_scale=2, norm_layer=LayerNorm))
            stages.append(VSSLayer(dim=skip, depth=2, attn_drop=0., drop_path=dpr[sum(depths[:s - 1]):sum(depths[:s])],
                                   d_state=math.ceil(2 * skip / 6) if d_state is None else d_state, norm_layer=LayerNorm,
                                   downsample=None, use_checkpoint=False))
            seg_layers.append(nn.Conv2d(skip, num_classes, 1, 1, 0, bias=True))
            concat_back_dim.append(TokenLinear(2 * skip, skip))
        expand_layers.append(FinalPatchExpand_X4(input_resolution=None, dim=ch[0], dim_scale=4, norm_layer=LayerNorm))
        stages.append(nn.Identity())
        seg_layers.append(nn.Conv2d(ch[0], num_classes, 1, 1, 0, bias=True))
        self.stages = nn.ModuleList(stages)
        self.expand_layers = nn.ModuleList(expand_layers)
        self.seg_layers = nn.ModuleList(seg_layers)
        self.concat_back_dim = nn.ModuleList(concat_back_dim)

    def forward(self, skips):
        lres = skips[-1]
        outs = []
        last = len(self.stages) - 1
        for s in range(len(self.stages)):
            x = self.expand_layers[s](lres)
            if s < last:
                x = self.concat_back_dim[s](torch.cat((x, skips[-(s + 2)].permute(0, 2, 3, 1)), -1))
            x = self.stages[s](x).permute(0, 3, 1, 2)
            if self.deep_supervision:
                outs.append(self.seg_layers[s](x))
            elif s == last:
                outs.append(self.seg_layers[-1](x))
            lres = x
        outs = outs[::-1]                      # largest prediction first
        return outs if self.deep_supervision else outs[0]


class _EncoderFreeze:
    """`freeze_encoder` / `unfreeze_encoder` of both networks (SwinUMamba.py:626-634): everything in the VSSM encoder except the
    patch embedding"""

    @torch.no_grad()
    def freeze_encoder(self):
        for name, param in self.vssm_encoder.named_parameters():
            if "patch_embed" not in name:
                param.requires_grad = False

    @torch.no_grad()
    def unfreeze_encoder(self):
        for param in self.vssm_encoder.parameters():
            param.requires_grad = True


class SwinUMambaD(_EncoderFreeze, nn.Module):
    """SwinUMambaD.py:641-661"""

    def __init__(self, vss_args, decoder_args):
        super().__init__()
        self.vssm_encoder = VSSMEncoder(**vss_args)
        self.decoder = UNetResDecoder(**decoder_args)

    @property
    def deep_supervision(self):
        return self.decoder.deep_supervision

    @deep_supervision.setter
    def deep_supervision(self, enabled):
        self.decoder.deep_supervision = enabled

    def forward(self, x):
        return self.decoder(self.vssm_encoder(x))


class SwinUMamba(_EncoderFreeze, nn.Module):
    """SwinUMamba.py:456-634: k7 s2 stem + affine InstanceNorm -> VSSMEncoder(patch 2) -> UNETR residual blocks on the input and
    on every encoder map -> five UnetrUpBlocks -> one more residual block -> 1x1 heads on the four finest decoder maps"""

    def __init__(self, in_chans=1, out_chans=13, feat_size=[48, 96, 192, 384, 768], drop_path_rate=0,
                 layer_scale_init_value=1e-6, hidden_size: int = 768, norm_name="instance", res_block: bool = True,
                 spatial_dims=2, deep_supervision: bool = False) -> None:
        super().__init__()
        self.hidden_size, self.in_chans, self.out_chans = hidden_size, in_chans, out_chans
        self.drop_path_rate, self.feat_size, self.layer_scale_init_value = drop_path_rate, feat_size, layer_scale_init_value
        f = feat_size
        self.stem = nn.Sequential(nn.Conv2d(in_chans, f[0], kernel_size=7, stride=2, padding=3),
                                  nn.InstanceNorm2d(f[0], eps=1e-5, affine=True))
        self.spatial_dims = spatial_dims
        self.vssm_encoder = VSSMEncoder(patch_size=2, in_chans=f[0])

        def basic(cin, cout):
            return UnetrBasicBlock(spatial_dims=spatial_dims, in_channels=cin, out_channels=cout, kernel_size=3, stride=1,
                                   norm_name=norm_name, res_block=res_block)

        def up(cin, cout):
            return UnetrUpBlock(spatial_dims=spatial_dims, in_channels=cin, out_channels=cout, kernel_size=3,
                                upsample_kernel_size=2, norm_name=norm_name, res_block=res_block)

        self.encoder1 = basic(in_chans, f[0])
        self.encoder2 = basic(f[0], f[1])
        self.encoder3 = basic(f[1], f[2])
        self.encoder4 = basic(f[2], f[3])
        self.encoder5 = basic(f[3], f[4])
        self.decoder6 = up(hidden_size, f[4])
        self.decoder5 = up(hidden_size, f[3])
        self.decoder4 = up(f[3], f[2])
        self.decoder3 = up(f[2], f[1])
        self.decoder2 = up(f[1], f[0])
        self.decoder1 = basic(f[0], f[0])
        self.deep_supervision = deep_supervision
        self.out_layers = nn.ModuleList([UnetOutBlock(spatial_dims=spatial_dims, in_channels=f[i], out_channels=out_chans)
                                         for i in range(4)])

    def forward(self, x_in):
        vss = self.vssm_encoder(self.stem(x_in))
        enc1 = self.encoder1(x_in)
        enc2 = self.encoder2(vss[0])
        enc3 = self.encoder3(vss[1])
        enc4 = self.encoder4(vss[2])
        enc5 = self.encoder5(vss[3])
        dec4 = self.decoder6(vss[4], enc5)
        dec3 = self.decoder5(dec4, enc4)
        dec2 = self.decoder4(dec3, enc3)
        dec1 = self.decoder3(dec2, enc2)
        dec0 = self.decoder2(dec1, enc1)
        dec_out = self.decoder1(dec0)
        if self.deep_supervision:
            return [self.out_layers[i](t) for i, t in enumerate((dec_out, dec1, dec2, dec3))]
        return self.out_layers[0](dec_out)


def load_pretrained_ckpt(model, ckpt_path="./data/pretrained/vmamba/vmamba_tiny_e292.pth", num_input_channels=None):
    """VMamba-tiny ImageNet weights into `vssm_encoder` (SwinUMamba.py:637-665, SwinUMambaD.py:664-694): classifier and final
    norm dropped, `layers.i.downsample` renamed to `downsamples.i`; the patch embedding is skipped always (Swin-UMamba) or when
    its input-channel count differs (`num_input_channels` given: the -D rule)."""
    print(f"Loading weights from: {ckpt_path}")
    ckpt = torch.load(ckpt_path, map_location='cpu')['model']
    skip = {"norm.weight", "norm.bias", "head.weight", "head.bias"}
    if num_input_channels is None:
        skip |= {"patch_embed.proj.weight", "patch_embed.proj.bias", "patch_embed.norm.weight"}
    model_dict = model.state_dict()
    for k, v in ckpt.items():
        if k in skip:
            continue
        if num_input_channels is not None and "patch_embed" in k \
                and ckpt["patch_embed.proj.weight"].shape[1] != num_input_channels:
            continue
        kr = re.sub(r"layers\.(\d+)\.downsample", r"downsamples.\1", f"vssm_encoder.{k}")
        if kr in model_dict:
            if v.shape != model_dict[kr].shape:
                raise ValueError(f"Shape mismatch: {k} {tuple(v.shape)} vs {tuple(model_dict[kr].shape)}")
            model_dict[kr] = v
    model.load_state_dict(model_dict)
    return model


def get_swin_umamba_from_plans(num_segmentation_heads: int, num_input_channels: int, deep_supervision: bool = True,
                               use_pretrain: bool = True):
    """SwinUMamba.py:668-683"""
    model = SwinUMamba(in_chans=num_input_channels, out_chans=num_segmentation_heads, deep_supervision=deep_supervision)
    if use_pretrain:
        model = load_pretrained_ckpt(model)
    return model


def get_swin_umamba_d_from_plans(plans_manager, dataset_json: dict, configuration_manager, num_input_channels: int,
                                 deep_supervision: bool = True, use_pretrain: bool = True):
    """SwinUMambaD.py:697-732 (the pretrained load is commented out there: `use_pretrain` is accepted and ignored)"""
    if configuration_manager is not None:
        ks = getattr(configuration_manager, "conv_kernel_sizes", None)      # (the trainers' view of the plans exposes patch_size)
        dim = len(ks[0]) if ks else len(configuration_manager.patch_size)
        assert dim == 2, "Only 2D supported at the moment"
    vss_args = dict(in_chans=num_input_channels, patch_size=4, dims=96, drop_path_rate=0.2)
    decoder_args = dict(num_classes=_m2._heads(plans_manager, dataset_json), deep_supervision=deep_supervision,
                        drop_path_rate=0.2, d_state=16)
    model = SwinUMambaD(vss_args, decoder_args)
    model.apply(InitWeights_He(1e-2))
    return model
