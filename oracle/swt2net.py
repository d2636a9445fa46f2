"""TEST INFRASTRUCTURE ONLY - CPU restatement (plain torch, fp32) of the reference's SwT2Net,
/root/reference/nnunetv2/nets/swt2net.py.  Only tests/ and bench.py's cpu_baseline leg import it; no product module does, and
nothing here touches the HIP library.

What is restated, with the reference lines it follows:
    REBNCONV :17-31 (depthwise 3x3 + pointwise 1x1, both without bias, -> BatchNorm -> ReLU; `dirate` unused), RSU4F :873-905
    DropPath :395-409, PatchEmbedding :412-432 (right / bottom padding by p - size % p on BOTH axes as soon as one does not
    divide), PatchMerging :435-464, PatchExpanding :467-478, FinalPatchExpanding :481-493, Mlp :496-515
    WindowAttention :518-619 - relative-position index :536-549, shift mask (-100 between regions) :559-582, forward :584-619:
        cyclic roll by -(7 // 2), 7x7 windows, softmax(q k^T * scale + bias [+ mask]) v, windows merged, roll back, proj
    SwinTransformerBlock :622-661 (padding on the TOP / LEFT by ws - size % ws on both axes as soon as one does not divide, crop
        from the end), BasicBlock / BasicBlockUp :664-740, SwinTransformerUnet :743-869, SwT2Net :909-1156
    macro-level PatchMerging2D / PatchExpand: oracle/m2net.py (same classes in both reference files)

Module / parameter names and registration order equal the reference's (tests/golden_util.det_fill, interchangeable
state_dicts).  PINNED by the reference's own outputs: tests/test_oracle_swt2net.py runs SwT2Net on
tests/golden/net_SwT2Net_64.npz (seven outputs of the reference's module, eval mode) and netgrad_SwT2Net_64.npz (its
autograd), and holds the state_dict keys to tests/golden/state_dict_manifest.json.
"""
from functools import partial

import torch
import torch.nn.functional as F
from torch import nn

from .m2net import PatchExpand, PatchMerging2D, _image, _tokens, _X2Net

WS = 7   # window edge of every Swin unit of the zoo


def _conv_only(in_ch, out_ch, k, groups=1, bias=True):
    """the `conv_only` form of monai's Convolution as the reference uses it: a Sequential whose only child is `conv`"""
    seq = nn.Sequential()
    seq.add_module("conv", nn.Conv2d(in_ch, out_ch, k, padding=(k - 1) // 2, groups=groups, bias=bias))
    return seq


def dw_separable(in_ch, out_ch):
    return nn.Sequential(_conv_only(in_ch, in_ch, 3, groups=in_ch, bias=False), _conv_only(in_ch, out_ch, 1, bias=False))


class REBNCONV(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv_s1 = dw_separable(in_ch, out_ch)
        self.bn_s1 = nn.BatchNorm2d(out_ch)

    def forward(self, x):
        return F.relu(self.bn_s1(self.conv_s1(x)))


class RSU4F(nn.Module):
    def __init__(self, in_ch, mid_ch, out_ch):
        super().__init__()
        self.rebnconvin = REBNCONV(in_ch, out_ch)
        self.rebnconv1 = REBNCONV(out_ch, mid_ch)
        self.rebnconv2 = REBNCONV(mid_ch, mid_ch)
        self.rebnconv3 = REBNCONV(mid_ch, mid_ch)
        self.rebnconv4 = REBNCONV(mid_ch, mid_ch)
        self.rebnconv3d = REBNCONV(2 * mid_ch, mid_ch)
        self.rebnconv2d = REBNCONV(2 * mid_ch, mid_ch)
        self.rebnconv1d = REBNCONV(2 * mid_ch, out_ch)

    def forward(self, x):
        x0 = self.rebnconvin(x)
        a = self.rebnconv1(x0)
        b = self.rebnconv2(a)
        c = self.rebnconv3(b)
        d = self.rebnconv4(c)
        d = self.rebnconv3d(torch.cat((d, c), 1))
        d = self.rebnconv2d(torch.cat((d, b), 1))
        d = self.rebnconv1d(torch.cat((d, a), 1))
        return d + x0


class DropPath(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if self.p == 0.0 or not self.training:
            return x
        keep = 1.0 - self.p
        m = (keep + torch.rand((x.shape[0],) + (1,) * (x.dim() - 1), dtype=x.dtype)).floor_()
        return x.div(keep) * m


def depth_to_space(x, s):
    B, H, W, C = x.shape
    return x.view(B, H, W, s, s, C // (s * s)).permute(0, 1, 3, 2, 4, 5).reshape(B, H * s, W * s, C // (s * s))


class PatchEmbedding(nn.Module):
    def __init__(self, patch, in_ch, dim, norm):
        super().__init__()
        self.patch = patch
        self.proj = nn.Conv2d(in_ch, dim, patch, patch)
        self.norm = norm(dim)

    def forward(self, x):
        p, (H, W) = self.patch, x.shape[2:]
        if H % p or W % p:
            x = F.pad(x, (0, p - W % p, 0, p - H % p))
        return self.norm(_tokens(self.proj(x)))


class PatchMerging(nn.Module):
    def __init__(self, dim, norm):
        super().__init__()
        self.norm = norm(4 * dim)
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)

    def forward(self, x):
        H, W = x.shape[1:3]
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat((x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]), -1)
        return self.reduction(self.norm(x))


class PatchExpanding(nn.Module):
    def __init__(self, dim, norm):
        super().__init__()
        self.expand = nn.Linear(dim, 2 * dim, bias=False)
        self.norm = norm(dim // 2)

    def forward(self, x):
        return self.norm(depth_to_space(self.expand(x), 2))


class FinalPatchExpanding(nn.Module):
    def __init__(self, dim, norm, patch):
        super().__init__()
        self.patch = patch
        self.expand = nn.Linear(dim, patch * patch * dim, bias=False)
        self.norm = norm(dim)

    def forward(self, x):
        return self.norm(depth_to_space(self.expand(x), self.patch))


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


def to_windows(x):     # (B, H, W, C) -> (B * nH * nW, 49, C)
    B, H, W, C = x.shape
    return x.view(B, H // WS, WS, W // WS, WS, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, WS * WS, C)


def from_windows(w, B, H, W):
    return w.view(B, H // WS, W // WS, WS, WS, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


class WindowAttention(nn.Module):
    def __init__(self, dim, heads, shift):
        super().__init__()
        self.heads, self.scale, self.shift = heads, (dim // heads) ** -0.5, WS // 2 if shift else 0
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * WS - 1) ** 2, heads))
        ar = torch.arange(WS)
        pos = torch.stack(torch.meshgrid(ar, ar, indexing="ij")).flatten(1)                 # (2, 49)
        rel = (pos[:, :, None] - pos[:, None, :]).permute(1, 2, 0) + (WS - 1)
        self.register_buffer("relative_position_index", rel[..., 0] * (2 * WS - 1) + rel[..., 1])
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def region_mask(self, H, W):
        """(nWindows, 49, 49): -100 where two tokens of a window of the rolled image come from different image regions"""
        region = torch.zeros(1, H, W, 1)
        spans = (slice(0, -WS), slice(-WS, -self.shift), slice(-self.shift, None))
        k = 0
        for hs in spans:
            for ws in spans:
                region[:, hs, ws, :] = k
                k += 1
        r = to_windows(region).squeeze(-1)
        return ((r[:, None, :] - r[:, :, None]) != 0).float() * -100.0

    def forward(self, x):          # (B, H, W, C), H and W multiples of 7
        B, H, W, C = x.shape
        if self.shift:
            x = torch.roll(x, (-self.shift, -self.shift), (1, 2))
        win = to_windows(x)
        nW = win.shape[0] // B
        q, k, v = self.qkv(win).view(-1, WS * WS, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        att = (q * self.scale) @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(WS * WS, WS * WS, -1)
        att = att + bias.permute(2, 0, 1)[None]
        if self.shift:
            att = (att.view(B, nW, self.heads, WS * WS, WS * WS) + self.region_mask(H, W)[None, :, None]).view(
                -1, self.heads, WS * WS, WS * WS)
        out = (att.softmax(-1) @ v).transpose(1, 2).reshape(-1, WS * WS, C)
        x = from_windows(out, B, H, W)
        if self.shift:
            x = torch.roll(x, (self.shift, self.shift), (1, 2))
        return self.proj(x)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, heads, shift, drop_path, norm):
        super().__init__()
        self.norm1 = norm(dim)
        self.attn = WindowAttention(dim, heads, shift)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm(dim)
        self.mlp = Mlp(dim, 4 * dim)

    def forward(self, x):
        H, W = x.shape[1:3]
        padded = H % WS != 0 or W % WS != 0
        if padded:
            x = F.pad(x, (0, 0, WS - W % WS, 0, WS - H % WS, 0))
        x = x + self.drop_path(self.attn(self.norm1(x)))
        x = x + self.drop_path(self.mlp(self.norm2(x)))
        return x[:, -H:, -W:, :] if padded else x


def stage_rates(depths, rate, index):
    r = torch.linspace(0, rate, sum(depths)).tolist()
    return r[sum(depths[:index]):sum(depths[:index + 1])]


def swin_blocks(dim, depth, heads, rates, norm):
    return nn.ModuleList(SwinTransformerBlock(dim, heads, bool(i % 2), rates[i], norm) for i in range(depth))


class BasicBlock(nn.Module):
    def __init__(self, index, dim0, depths, heads, rate, norm, merge):
        super().__init__()
        dim = dim0 * 2 ** index
        self.blocks = swin_blocks(dim, depths[index], heads[index], stage_rates(depths, rate, index), norm)
        self.downsample = PatchMerging(dim, norm) if merge else None

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return self.downsample(x) if self.downsample is not None else x


class BasicBlockUp(nn.Module):
    def __init__(self, index, dim0, depths, heads, rate, norm, expand):
        super().__init__()
        index = len(depths) - index - 2
        dim = dim0 * 2 ** index
        self.blocks = swin_blocks(dim, depths[index], heads[index], stage_rates(depths, rate, index), norm)
        self.upsample = PatchExpanding(dim, norm) if expand else nn.Identity()

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return self.upsample(x)


class SwinTransformerUnet(nn.Module):
    """as SwT2Net builds it: depths (2, 2, 4, 2), drop path 0.1, a depthwise-separable stem whose output is added at the end"""

    def __init__(self, patch, in_ch, out_ch, dim, heads, depths=(2, 2, 4, 2), rate=0.1):
        super().__init__()
        norm = partial(nn.LayerNorm, eps=1e-6)
        n = len(depths)
        self.rebnconvin = dw_separable(in_ch, out_ch)
        self.patch_embed = PatchEmbedding(patch, in_ch, dim, norm)
        self.layers = nn.ModuleList(BasicBlock(i, dim, depths, heads, rate, norm, i != n - 1) for i in range(n))
        self.first_patch_expanding = PatchExpanding(dim * 2 ** (n - 1), norm)
        self.layers_up = nn.ModuleList(BasicBlockUp(i, dim, depths, heads, rate, norm, i < n - 2) for i in range(n - 1))
        self.skip_connection_layers = nn.ModuleList(
            nn.Linear(dim * 2 ** (n - 2 - i) * 2, dim * 2 ** (n - 2 - i)) for i in range(n - 1))
        self.norm_up = norm(dim)
        self.final_patch_expanding = FinalPatchExpanding(dim, norm, patch)
        self.head = nn.Conv2d(dim, out_ch, 1, bias=False)

    def forward(self, x):
        stem = self.rebnconvin(x)
        x = self.patch_embed(x)
        inputs = []
        for layer in self.layers:
            inputs.append(x)
            x = layer(x)
        x = self.first_patch_expanding(x)
        for i, layer in enumerate(self.layers_up):
            skip = inputs[len(inputs) - i - 2]
            x = x[:, :skip.shape[1], :skip.shape[2]]          # rows / columns that came from odd-size padding
            x = layer(self.skip_connection_layers[i](torch.cat((x, skip), -1)))
        x = self.final_patch_expanding(self.norm_up(x))
        return self.head(_image(x)) + stem


class SwT2Net(_X2Net):
    def __init__(self, in_ch, out_ch, deep_supervision=True):
        super().__init__()
        self.deep_supervision = deep_supervision
        unit = {1: (4, 32, 32, (2, 2, 4, 8)), 2: (4, 64, 64, (2, 4, 8, 16)), 3: (2, 128, 96, (3, 6, 12, 24)),
                4: (1, 256, 96, (3, 6, 12, 24))}     # stage: patch size, output channels, embedding width, heads

        def su(k, i):
            p, o, e, h = unit[k]
            return SwinTransformerUnet(p, i, o, e, h)

        self.stage1 = su(1, in_ch)
        self.patch_merging1 = PatchMerging2D(32)
        self.stage2 = su(2, 64)
        self.patch_merging2 = PatchMerging2D(64)
        self.stage3 = su(3, 128)
        self.patch_merging3 = PatchMerging2D(128)
        self.stage4 = su(4, 256)
        self.patch_merging4 = PatchMerging2D(256)
        self.stage5 = RSU4F(512, 256, 512)
        self.pool56 = nn.MaxPool2d(2, stride=2, ceil_mode=True)
        self.stage6 = RSU4F(512, 256, 512)
        self.stage5d = RSU4F(1024, 256, 512)
        for k, c in ((4, 512), (3, 256), (2, 128), (1, 64)):
            setattr(self, f"patch_expand{k}d", PatchExpand(c, 2))
            setattr(self, f"concat_back_dim{k}d", nn.Linear(c, c // 2))
            setattr(self, f"stage{k}d", su(k, c // 2))
        for i, c in enumerate((32, 64, 128, 256, 512, 512), 1):
            setattr(self, f"side{i}", _conv_only(c, out_ch, 1))
        self.outconv = _conv_only(6 * out_ch, out_ch, 1)

    def fuse(self, k, up_tokens, skip):
        return _image(getattr(self, f"concat_back_dim{k}d")(torch.cat((up_tokens, _tokens(skip)), -1)))
