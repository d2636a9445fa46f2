"""U^2-Net / U^2-Net-P (reference: /root/reference/nnunetv2/nets/u2net.py:10-24 REBNCONV, :33-318 RSU7 / RSU6 / RSU5 / RSU4 /
RSU4F, :320-417 U2NET, :440-537 U2NETP, factories :560-594; trainers nnUNetTrainerU2Net[P]).

Same modules, parameter names, registration order (seeded construction draws the reference's RNG stream) and forward
arithmetic; what differs is where the conv -> BatchNorm -> ReLU unit runs: under the fp16 autocast step of the trainer a
REBNCONV whose channel counts are multiples of 32 executes on the tap-table MFMA conv kernels with batch statistics from
the conv epilogue (nnuzoo_amd/rebnconv.py, csrc/conv_fprop.hip) on channels-last fp16 tensors.  Between units the tensors
stay in channels-last memory format (NCHW views of NHWC storage), which max_pool2d, interpolate and cat keep - so the
per-unit dispatch costs no layout copies.  The reference's RSU7 ... RSU4 repeat one pattern with 7 ... 4 levels: one
table-driven class `_RSU(levels)` builds them (attributes `rebnconvin`, `rebnconv1..L`, `pool1..L-2`, `rebnconv{L-1}d..1d`
in the reference's order)."""
from __future__ import annotations

import torch

from .. import backends as _backends
import torch.nn.functional as F
from torch import nn

from .. import rebnconv as _rb
from ..utilities.network_initialization import InitWeights_He
from .common2d import RSU4F  # noqa: F401  (same class as m2net's; re-exported under the reference's name)


class REBNCONV(nn.Module):
    """u2net.py:10-24"""
    backend = "unset"

    def __init__(self, in_ch=3, out_ch=3, dirate=1):
        super().__init__()
        self.conv_s1 = nn.Conv2d(in_ch, out_ch, 3, padding=1 * dirate, dilation=1 * dirate)
        self.bn_s1 = nn.BatchNorm2d(out_ch)
        self.relu_s1 = nn.ReLU(inplace=True)

    def _hip_ok(self, x: torch.Tensor) -> bool:
        if not (_rb.USE_HIP and x.is_cuda and x.dim() == 4 and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.float16 and _rb.supported(self.conv_s1, self.bn_s1)):
            return False
        if not self.bn_s1.training and torch.is_grad_enabled() and \
                (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return False                      # eval-mode statistics with autograd: torch path
        return True

    def forward(self, x):
        if self._hip_ok(x):
            _backends.note(self, "hip")
            xc = x.permute(0, 2, 3, 1)        # free when x is already channels-last in memory
            if xc.dtype != torch.float16 or not xc.is_contiguous():
                xc = xc.to(torch.float16).contiguous()
            return _rb.rebnconv_cl(self, xc).permute(0, 3, 1, 2)
        _backends.note(self, "library", why="outside fp16 autocast / unsupported channels / eval with autograd")
        return self.relu_s1(self.bn_s1(self.conv_s1(x)))


def _upsample_like(src, tar):
    """u2net.py:26-30"""
    return F.interpolate(src, size=tar.shape[2:], mode='bilinear')


class _RSU(nn.Module):
    """residual U block with L levels (u2net.py:33-282): encoder units 1..L-1 with ceil-mode 2x2 max pooling between
    them, a dilation-2 unit at the bottom, decoder units (L-1)d..1d on [upsampled, skip] concatenations, + input unit"""
    LEVELS = 7

    def __init__(self, in_ch=3, mid_ch=12, out_ch=3):
        super().__init__()
        L = self.LEVELS
        self.rebnconvin = REBNCONV(in_ch, out_ch, dirate=1)
        for i in range(1, L):
            setattr(self, f"rebnconv{i}", REBNCONV(out_ch if i == 1 else mid_ch, mid_ch, dirate=1))
            if i < L - 1:
                setattr(self, f"pool{i}", nn.MaxPool2d(2, stride=2, ceil_mode=True))
        setattr(self, f"rebnconv{L}", REBNCONV(mid_ch, mid_ch, dirate=2))
        for i in range(L - 1, 0, -1):
            setattr(self, f"rebnconv{i}d", REBNCONV(mid_ch * 2, out_ch if i == 1 else mid_ch, dirate=1))

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


class RSU6(_RSU):
    LEVELS = 6


class RSU5(_RSU):
    LEVELS = 5


class RSU4(_RSU):
    LEVELS = 4


class _U2(nn.Module):
    """U2NET (u2net.py:320-417) / U2NETP (:440-537): six encoder stages, five decoder stages, six side outputs fused by a
    1x1 conv; `CFG` = (mid, out) channels per encoder stage + decoder (in, mid, out)"""
    ENC = ()
    DEC = ()
    SIDE = ()

    def __init__(self, in_ch=3, out_ch=1, deep_supervision=False):
        super().__init__()
        self.deep_supervision = deep_supervision
        blocks = [RSU7, RSU6, RSU5, RSU4, RSU4F, RSU4F]
        cin = in_ch
        for s, (blk, (mid, cout)) in enumerate(zip(blocks, self.ENC), start=1):
            setattr(self, f"stage{s}", blk(cin, mid, cout))
            if s < 6:
                setattr(self, f"pool{s}{s + 1}", nn.MaxPool2d(2, stride=2, ceil_mode=True))
            cin = cout
        for s, blk, (ci, mid, co) in zip((5, 4, 3, 2, 1), (RSU4F, RSU4, RSU5, RSU6, RSU7), self.DEC):
            setattr(self, f"stage{s}d", blk(ci, mid, co))
        for s, c in enumerate(self.SIDE, start=1):
            setattr(self, f"side{s}", nn.Conv2d(c, out_ch, 3, padding=1))
        self.outconv = nn.Conv2d(6 * out_ch, out_ch, 1)

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


def get_u2netp_from_plans(num_segmentation_heads: int, num_input_channels: int, deep_supervision: bool = True,
                          use_pretrain: bool = True):
    """u2net.py:560-575 (He init with slope 1e-2; init_last_bn_before_add_to_0 has nothing to act on in these nets)"""
    model = U2NETP(in_ch=num_input_channels, out_ch=num_segmentation_heads, deep_supervision=deep_supervision)
    model.apply(InitWeights_He(1e-2))
    return model


def get_u2net_from_plans(num_segmentation_heads: int, num_input_channels: int, deep_supervision: bool = True,
                         use_pretrain: bool = True):
    """u2net.py:578-594"""
    model = U2NET(in_ch=num_input_channels, out_ch=num_segmentation_heads, deep_supervision=deep_supervision)
    model.apply(InitWeights_He(1e-2))
    return model
