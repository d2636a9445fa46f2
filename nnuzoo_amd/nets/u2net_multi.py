"""N-D U^2-Net / U^2-Net-P on monai's `Convolution` unit (reference: /root/reference/nnunetv2/nets/u2net_multi.py:56-340 RSU7 / RSU6 /
RSU5 / RSU4 / RSU4F, :343-462 U2NET, :465-645 U2NETP, factories :648-721; trainers nnUNetTrainerU2NetMulti[P]).

What the reference builds: every unit is `monai.networks.blocks.convolutions.Convolution(spatial_dims, in, out, dilation=d[, act,
norm])` - conv k3 (same padding, bias) followed by monai's ADN block in "NDA" order.  RSU6 / RSU5 / RSU4 / RSU4F pass
act="relu", norm="BATCH" (conv -> BatchNorm -> ReLU: the REBNCONV unit of nets/u2net.py under monai's child names); RSU7 does NOT
forward the two arguments (u2net_multi.py:61-87), so its units run monai's defaults: InstanceNorm (no affine) -> PReLU(0.25).
The six side convolutions of U2NET are conv_only (k3), those of U2NETP are full default units (:514-519, a quirk: InstanceNorm +
PReLU on the logits), `outconv` is a conv_only 1x1.  Pooling is MaxPool{2,3}d(2, ceil_mode) and `_upsample_like` is monai's
non-trainable UpSample = nn.Upsample(size, bi- / trilinear, align_corners=False).

PARITY UNPINNED: monai is absent from /root/reference and from this image.  `Convolution` below restates the published
monai 1.3 block (child names `conv`, `adn.N`, `adn.A`; PReLU with one parameter initialised to 0.25; InstanceNorm without affine
or running statistics) - the same restatement tools/ref_shim.py substitutes when it imports the reference's module to write
tests/golden/u2net_multi_*.npz, so the fixtures pin the reference's WIRING (stage / unit order, channel plan, pooling, up-sampling,
skip sums, the mae-off forward) and this file's arithmetic against torch, not monai's internals.

Where it runs: 2-D conv -> BatchNorm -> ReLU units with channel counts that are multiples of 32 (U2NET's RSU6 ... RSU4F) execute
under the trainer's fp16 autocast step on the tap-table MFMA conv kernels with batch statistics from the conv epilogue
(nnuzoo_amd/rebnconv.py, csrc/conv_fprop.hip) - the unit nets/u2net.py uses; every other unit (RSU7's InstanceNorm + PReLU, 3-D,
U2NETP's 16-channel plan) is stock torch on the device and recorded as such (nnuzoo_amd/backends.py).
The masked-autoencoder branch of U2NETP (`mae=True`, :476-483, :560-600; used by no trainer) is not built."""
from __future__ import annotations

import torch
from torch import nn

from .. import backends as _backends
from .. import rebnconv as _rb
from ..utilities.network_initialization import InitWeights_He


class MaxPool(nn.Module):
    """u2net_multi.py:14-36"""

    def __init__(self, spatial_dims: int, kernel_size, stride=None, padding=0, dilation=1, return_indices: bool = False,
                 ceil_mode: bool = False):
        super().__init__()
        cls = {2: nn.MaxPool2d, 3: nn.MaxPool3d}[spatial_dims]
        self.max_pool = cls(kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation,
                            return_indices=return_indices, ceil_mode=ceil_mode)

    def forward(self, input):
        return self.max_pool(input)


def _upsample_like(src, tar, upsample_mode="nontrainable"):
    """u2net_multi.py:40-52: monai UpSample(mode=nontrainable, interp LINEAR, align_corners False, size = target's)"""
    nd = src.dim() - 2
    return nn.functional.interpolate(src, size=tuple(tar.shape[2:]), mode={2: "bilinear", 3: "trilinear"}[nd], align_corners=False)


class ADN(nn.Sequential):
    """monai ADN in its default "NDA" order without dropout: children `N` (norm) and `A` (activation)"""

    def __init__(self, spatial_dims: int, channels: int, act: str, norm: str):
        super().__init__()
        n, a = norm.upper(), act.upper()
        if n == "INSTANCE":
            self.add_module("N", {2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}[spatial_dims](channels))
        elif n == "BATCH":
            self.add_module("N", {2: nn.BatchNorm2d, 3: nn.BatchNorm3d}[spatial_dims](channels))
        else:
            raise NotImplementedError(f"norm {norm!r}: the reference's u2net_multi uses INSTANCE and BATCH only")
        if a == "PRELU":
            self.add_module("A", nn.PReLU())
        elif a == "RELU":
            self.add_module("A", nn.ReLU())
        else:
            raise NotImplementedError(f"act {act!r}: the reference's u2net_multi uses PRELU and relu only")


class Convolution(nn.Sequential):
    """monai Convolution as u2net_multi.py calls it: conv (same padding for the dilation, bias) [+ ADN]"""
    backend = "unset"

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, strides=1, kernel_size=3, act="PRELU",
                 norm="INSTANCE", dilation=1, bias=True, conv_only=False, padding=None):
        super().__init__()
        pad = (kernel_size - 1) // 2 * dilation if padding is None else padding
        conv = {2: nn.Conv2d, 3: nn.Conv3d}[spatial_dims]
        self.add_module("conv", conv(in_channels, out_channels, kernel_size, strides, pad, dilation, 1, bias))
        if not conv_only:
            self.add_module("adn", ADN(spatial_dims, out_channels, act, norm))

    def hip_capable(self) -> bool:
        """the unit's STRUCTURE fits the MFMA conv path: conv k3 (dilation 1 / 2 / 4 / 8, channels in multiples of 32) -> BatchNorm2d ->
        ReLU.  (param_shadow.py leaves the fp32 master weights of such units alone: rebnconv.py packs them itself.)"""
        adn = self._modules.get("adn")
        return adn is not None and isinstance(adn.N, nn.BatchNorm2d) and isinstance(adn.A, nn.ReLU) \
            and isinstance(self.conv, nn.Conv2d) and _rb.supported(self.conv, adn.N)

    def _hip_unit(self, x: torch.Tensor) -> bool:
        """conv -> BatchNorm2d -> ReLU on the MFMA conv kernels (rebnconv.py) - same conditions as nets/u2net.py REBNCONV"""
        if not self.hip_capable():
            return False
        adn = self.adn
        if not (_rb.USE_HIP and x.is_cuda and x.dim() == 4 and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.float16):
            return False
        if not adn.N.training and torch.is_grad_enabled() and \
                (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return False
        return True

    # rebnconv.rebnconv_cl reads the unit through the REBNCONV names
    @property
    def conv_s1(self):
        return self.conv

    @property
    def bn_s1(self):
        return self.adn.N

    def forward(self, x):
        if self._hip_unit(x):
            _backends.note(self, "hip")
            xc = x.permute(0, 2, 3, 1)
            if xc.dtype != torch.float16 or not xc.is_contiguous():
                xc = xc.to(torch.float16).contiguous()
            return _rb.rebnconv_cl(self, xc).permute(0, 3, 1, 2)
        if "adn" in self._modules:
            _backends.note(self, "library", why="InstanceNorm + PReLU unit / 3-D / channels not multiples of 32 / outside fp16 autocast")
        return super().forward(x)


class _RSU(nn.Module):
    """residual U block with L levels (u2net_multi.py:56-304); registration order = the reference's"""
    LEVELS = 7
    FORWARDS_ACT_NORM = True      # RSU7 drops the act / norm arguments (:61-87): its units are InstanceNorm + PReLU

    def __init__(self, spatial_dims: int = 2, in_ch=3, mid_ch=12, out_ch=3, act="relu", norm="BATCH"):
        super().__init__()
        L = self.LEVELS
        kw = dict(act=act, norm=norm) if self.FORWARDS_ACT_NORM else {}
        self.rebnconvin = Convolution(spatial_dims, in_ch, out_ch, dilation=1, **kw)
        for i in range(1, L):
            setattr(self, f"rebnconv{i}", Convolution(spatial_dims, out_ch if i == 1 else mid_ch, mid_ch, dilation=1, **kw))
            if i < L - 1:
                setattr(self, f"pool{i}", MaxPool(spatial_dims, 2, stride=2, ceil_mode=True))
        setattr(self, f"rebnconv{L}", Convolution(spatial_dims, mid_ch, mid_ch, dilation=2, **kw))
        for i in range(L - 1, 0, -1):
            setattr(self, f"rebnconv{i}d", Convolution(spatial_dims, mid_ch * 2, out_ch if i == 1 else mid_ch, dilation=1, **kw))

    def forward(self, x):
        L = self.LEVELS
        hxin = self.rebnconvin(x)
        skips, hx = [], hxin
        for i in range(1, L):
            hx = getattr(self, f"rebnconv{i}")(hx)
            skips.append(hx)
            if i < L - 1:
                hx = getattr(self, f"pool{i}")(hx)
        hx = getattr(self, f"rebnconv{L}")(hx)
        for i in range(L - 1, 0, -1):
            hx = getattr(self, f"rebnconv{i}d")(torch.cat((hx, skips[i - 1]), 1))
            if i > 1:
                hx = _upsample_like(hx, skips[i - 2])
        return hx + hxin


class RSU7(_RSU):
    LEVELS = 7
    FORWARDS_ACT_NORM = False

    def __init__(self, spatial_dims: int = 2, in_ch=3, mid_ch=12, out_ch=3):
        super().__init__(spatial_dims, in_ch, mid_ch, out_ch)


class RSU6(_RSU):
    LEVELS = 6


class RSU5(_RSU):
    LEVELS = 5


class RSU4(_RSU):
    LEVELS = 4


class RSU4F(nn.Module):
    """u2net_multi.py:307-340: dilations 1, 2, 4, 8 and back, no pooling"""

    def __init__(self, spatial_dims: int = 2, in_ch=3, mid_ch=12, out_ch=3, act="relu", norm="BATCH"):
        super().__init__()
        kw = dict(act=act, norm=norm)
        self.rebnconvin = Convolution(spatial_dims, in_ch, out_ch, dilation=1, **kw)
        self.rebnconv1 = Convolution(spatial_dims, out_ch, mid_ch, dilation=1, **kw)
        self.rebnconv2 = Convolution(spatial_dims, mid_ch, mid_ch, dilation=2, **kw)
        self.rebnconv3 = Convolution(spatial_dims, mid_ch, mid_ch, dilation=4, **kw)
        self.rebnconv4 = Convolution(spatial_dims, mid_ch, mid_ch, dilation=8, **kw)
        self.rebnconv3d = Convolution(spatial_dims, mid_ch * 2, mid_ch, dilation=4, **kw)
        self.rebnconv2d = Convolution(spatial_dims, mid_ch * 2, mid_ch, dilation=2, **kw)
        self.rebnconv1d = Convolution(spatial_dims, mid_ch * 2, out_ch, dilation=1, **kw)

    def forward(self, x):
        hxin = self.rebnconvin(x)
        hx1 = self.rebnconv1(hxin)
        hx2 = self.rebnconv2(hx1)
        hx3 = self.rebnconv3(hx2)
        hx4 = self.rebnconv4(hx3)
        hx3d = self.rebnconv3d(torch.cat((hx4, hx3), 1))
        hx2d = self.rebnconv2d(torch.cat((hx3d, hx2), 1))
        hx1d = self.rebnconv1d(torch.cat((hx2d, hx1), 1))
        return hx1d + hxin


class _U2(nn.Module):
    """U2NET (:343-462) / U2NETP (:465-645, mae off): six encoder stages, five decoder stages, six side outputs, 1x1 fuse"""
    ENC = ()
    DEC = ()
    SIDE = ()
    SIDE_CONV_ONLY = True

    def __init__(self, spatial_dims: int = 2, in_ch=3, out_ch=1, deep_supervision=False):
        super().__init__()
        self.spatial_dims = spatial_dims
        self.deep_supervision = deep_supervision
        blocks = [RSU7, RSU6, RSU5, RSU4, RSU4F, RSU4F]
        cin = in_ch
        for s, (blk, (mid, cout)) in enumerate(zip(blocks, self.ENC), start=1):
            setattr(self, f"stage{s}", blk(spatial_dims, cin, mid, cout))
            if s < 6:
                setattr(self, f"pool{s}{s + 1}", MaxPool(spatial_dims, 2, stride=2, ceil_mode=True))
            cin = cout
        for s, blk, (ci, mid, co) in zip((5, 4, 3, 2, 1), (RSU4F, RSU4, RSU5, RSU6, RSU7), self.DEC):
            setattr(self, f"stage{s}d", blk(spatial_dims, ci, mid, co))
        for s, c in enumerate(self.SIDE, start=1):
            setattr(self, f"side{s}", Convolution(spatial_dims, c, out_ch, kernel_size=3, padding=1,
                                                  conv_only=self.SIDE_CONV_ONLY))
        self.outconv = Convolution(spatial_dims, 6 * out_ch, out_ch, kernel_size=1, conv_only=True)

    def forward(self, x):
        hx, enc = x, []
        for s in range(1, 7):
            h = getattr(self, f"stage{s}")(hx)
            enc.append(h)
            if s < 6:
                hx = getattr(self, f"pool{s}{s + 1}")(h)
        hx6 = enc[5]
        dec = {}
        up = _upsample_like(hx6, enc[4])
        for s in (5, 4, 3, 2, 1):
            d = getattr(self, f"stage{s}d")(torch.cat((up, enc[s - 1]), 1))
            dec[s] = d
            if s > 1:
                up = _upsample_like(d, enc[s - 2])
        d1 = self.side1(dec[1])
        sides = [d1] + [_upsample_like(getattr(self, f"side{s}")(dec[s]), d1) for s in (2, 3, 4, 5)]
        sides.append(_upsample_like(self.side6(hx6), d1))
        d0 = self.outconv(torch.cat(sides, 1))
        if self.deep_supervision:
            return (d0, *sides)
        return d0

    def _encoder_groups(self):
        return [getattr(self, f"stage{s}") for s in range(1, 7)]

    @torch.no_grad()
    def freeze_encoder(self):
        for group in self._encoder_groups():
            for p in group.parameters():
                p.requires_grad = False

    @torch.no_grad()
    def unfreeze_encoder(self):
        for group in self._encoder_groups():
            for p in group.parameters():
                p.requires_grad = True


class U2NET(_U2):
    ENC = ((32, 64), (32, 128), (64, 256), (128, 512), (256, 512), (256, 512))
    DEC = ((1024, 256, 512), (1024, 128, 256), (512, 64, 128), (256, 32, 64), (128, 16, 64))
    SIDE = (64, 64, 128, 256, 512, 512)


class U2NETP(_U2):
    ENC = ((16, 64),) * 6
    DEC = ((128, 16, 64),) * 5
    SIDE = (64,) * 6
    SIDE_CONV_ONLY = False       # :514-519: the side units keep monai's default InstanceNorm + PReLU

    def __init__(self, spatial_dims: int = 2, in_ch=3, out_ch=1, deep_supervision=False, mae=False, mask_ratio=0.75):
        if mae:
            raise NotImplementedError("U2NETP(mae=True): the masked-autoencoder branch (u2net_multi.py:476-483, 560-600) is used by "
                                      "no trainer and is not built")
        super().__init__(spatial_dims, in_ch, out_ch, deep_supervision)
        self.in_ch = in_ch
        self.mae = False


def get_u2netp_from_plans(plans_manager, dataset_json: dict, configuration_manager, num_input_channels: int,
                          deep_supervision: bool = True, use_pretrain: bool = True):
    """u2net_multi.py:648-669 (the plans-style signature; dimensionality = len(patch_size))"""
    from .m2net import _heads
    model = U2NETP(spatial_dims=len(configuration_manager.patch_size), in_ch=num_input_channels,
                   out_ch=_heads(plans_manager, dataset_json), deep_supervision=deep_supervision)
    model.apply(InitWeights_He(1e-2))
    return model


def get_u2net_from_plans(spatial_dims: int, num_segmentation_heads: int, num_input_channels: int,
                         deep_supervision: bool = True, use_pretrain: bool = True):
    """u2net_multi.py:699-721 (NOT the plans-style signature: the reference's own trainer calls it with the plans-style arguments
    and fails, nnUNetTrainerU2NetMulti.py:37-44)"""
    model = U2NET(spatial_dims=spatial_dims, in_ch=num_input_channels, out_ch=num_segmentation_heads,
                  deep_supervision=deep_supervision)
    model.apply(InitWeights_He(1e-2))
    return model
