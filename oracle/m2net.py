"""TEST INFRASTRUCTURE ONLY - CPU restatement (plain torch, fp32) of the reference's M2Net / M2NetP ("SS2D^2Net"),
/root/reference/nnunetv2/nets/m2net.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it; no
product module does, and nothing here touches the HIP library.

What is restated, with the reference lines it follows:
    REBNCONV :18-30, _upsample_like :33-36 (bilinear, align_corners=False), RSU4F :769-801
    SS2D :39-225 - forward_core :170-206 (four scan orders: rows, columns and their reversals; x_proj / dt_proj einsums;
        selective scan in fp32 with delta softplus + bias; the reversed and transposed results brought back to row-major
        order), forward :208-225 (in_proj -> split x, z -> depthwise conv + SiLU -> core -> sum of the four -> LayerNorm
        -> * SiLU(z) -> out_proj)
    PatchMerging2D :228-273 (2x2 space-to-depth in the order (0,0), (1,0), (0,1), (1,1) -> LayerNorm -> Linear)
    PatchExpand :276-319 (Linear + depth-to-space, or depth-to-space + Linear when output_dim is given; LayerNorm last)
    VSSMDecoder :359-483, PatchEmbed2D :486-510, VSSBlock / VSSLayer :513-595, VSSMEncoder :598-710, MU :713-765
    M2Net :805-971, M2NetP :1011-1184 (macro wiring :883-956)
The selective scan is oracle/selective_scan.py (the reference's selective_scan_ref semantics; the CUDA extension the
reference imports at :11 is absent).

Module / parameter names and registration order equal the reference's, so that tests/golden_util.det_fill gives both the
same parameters and state_dicts interchange.  PINNED by the reference's own outputs: tests/test_oracle_m2net.py runs these
classes on tests/golden/net_M2NetP_64.npz, net_M2Net_64.npz (seven outputs of the reference's modules, eval mode) and
netgrad_M2NetP_64.npz, netgrad_M2Net_64.npz (their autograd: dx, samples and norms of every parameter gradient), and holds the
state_dict keys to
tests/golden/state_dict_manifest.json.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .selective_scan import selective_scan_torch


def _tokens(x):   # (B, C, H, W) -> (B, H, W, C)
    return x.permute(0, 2, 3, 1)


def _image(x):    # (B, H, W, C) -> (B, C, H, W)
    return x.permute(0, 3, 1, 2)


def upsample_like(src, size):
    return F.interpolate(src, size=tuple(size), mode="bilinear", align_corners=False)


class StochasticDepth(nn.Module):
    """timm DropPath as the reference uses it: one Bernoulli draw per sample, kept samples scaled by 1 / keep"""

    def __init__(self, p: float):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if self.p == 0.0 or not self.training:
            return x
        keep = 1.0 - self.p
        m = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * (m / keep)


class REBNCONV(nn.Module):
    def __init__(self, in_ch, out_ch, dirate=1):
        super().__init__()
        self.conv_s1 = nn.Conv2d(in_ch, out_ch, 3, padding=dirate, dilation=dirate)
        self.bn_s1 = nn.BatchNorm2d(out_ch)

    def forward(self, x):
        return F.relu(self.bn_s1(self.conv_s1(x)))


class RSU4F(nn.Module):
    def __init__(self, in_ch, mid_ch, out_ch):
        super().__init__()
        self.rebnconvin = REBNCONV(in_ch, out_ch, 1)
        self.rebnconv1 = REBNCONV(out_ch, mid_ch, 1)
        self.rebnconv2 = REBNCONV(mid_ch, mid_ch, 2)
        self.rebnconv3 = REBNCONV(mid_ch, mid_ch, 4)
        self.rebnconv4 = REBNCONV(mid_ch, mid_ch, 8)
        self.rebnconv3d = REBNCONV(2 * mid_ch, mid_ch, 4)
        self.rebnconv2d = REBNCONV(2 * mid_ch, mid_ch, 2)
        self.rebnconv1d = REBNCONV(2 * mid_ch, out_ch, 1)

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


class SS2D(nn.Module):
    def __init__(self, d_model, d_state=16, expand=2):
        super().__init__()
        Di, N, R = expand * d_model, d_state, math.ceil(d_model / 16)
        self.Di, self.N, self.R = Di, N, R
        self.in_proj = nn.Linear(d_model, 2 * Di, bias=False)
        self.conv2d = nn.Conv2d(Di, Di, 3, padding=1, groups=Di, bias=True)
        self.x_proj_weight = nn.Parameter(torch.randn(4, R + 2 * N, Di) * Di ** -0.5)
        self.dt_projs_weight = nn.Parameter((torch.rand(4, Di, R) * 2 - 1) * R ** -0.5)
        dt = torch.exp(torch.rand(4, Di) * (math.log(0.1) - math.log(0.001)) + math.log(0.001)).clamp(min=1e-4)
        self.dt_projs_bias = nn.Parameter(dt + torch.log(-torch.expm1(-dt)))
        self.A_logs = nn.Parameter(torch.log(torch.arange(1, N + 1, dtype=torch.float32)).repeat(4 * Di, 1))
        self.Ds = nn.Parameter(torch.ones(4 * Di))
        self.out_norm = nn.LayerNorm(Di)
        self.out_proj = nn.Linear(Di, d_model, bias=False)

    def four_scans(self, x):
        """x (B, Di, H, W) -> sum over the four scan orders, (B, Di, H * W) in row-major order"""
        B, Di, H, W = x.shape
        L, N, R = H * W, self.N, self.R
        by_rows = x.flatten(2)
        by_cols = x.transpose(2, 3).flatten(2)
        seq = torch.stack((by_rows, by_cols, by_rows.flip(-1), by_cols.flip(-1)), 1)          # (B, 4, Di, L)
        proj = torch.einsum("bkdl,kcd->bkcl", seq, self.x_proj_weight)
        dt, Bm, Cm = proj.split((R, N, N), 2)
        dt = torch.einsum("bkrl,kdr->bkdl", dt, self.dt_projs_weight)
        y = selective_scan_torch(seq.reshape(B, 4 * Di, L), dt.reshape(B, 4 * Di, L), -torch.exp(self.A_logs.float()),
                                 Bm, Cm, self.Ds.float(), self.dt_projs_bias.reshape(-1).float(), True).view(B, 4, Di, L)

        def cols_to_rows(t):
            return t.reshape(B, Di, W, H).transpose(2, 3).reshape(B, Di, L)

        return y[:, 0] + y[:, 2].flip(-1) + cols_to_rows(y[:, 1]) + cols_to_rows(y[:, 3].flip(-1))

    def forward(self, x):          # (B, H, W, C)
        B, H, W, _ = x.shape
        x, z = self.in_proj(x).chunk(2, -1)
        x = F.silu(self.conv2d(_image(x)))
        y = self.four_scans(x).transpose(1, 2).reshape(B, H, W, -1)
        return self.out_proj(self.out_norm(y) * F.silu(z))


class VSSBlock(nn.Module):
    def __init__(self, dim, drop_path=0.0):
        super().__init__()
        self.ln_1 = nn.LayerNorm(dim)
        self.self_attention = SS2D(dim)
        self.drop_path = StochasticDepth(drop_path)

    def forward(self, x):
        return x + self.drop_path(self.self_attention(self.ln_1(x)))


class VSSLayer(nn.Module):
    def __init__(self, dim, depth, drop_path):
        super().__init__()
        self.blocks = nn.ModuleList(VSSBlock(dim, drop_path[i]) for i in range(depth))

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


class PatchEmbed2D(nn.Module):
    def __init__(self, patch, in_ch, dim):
        super().__init__()
        self.proj = nn.Conv2d(in_ch, dim, patch, patch)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        return self.norm(_tokens(self.proj(x)))


class PatchMerging2D(nn.Module):
    def __init__(self, dim, out=None):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, out or 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x, image=False):
        if image:
            x = _tokens(x)
        H2, W2 = x.shape[1] // 2 * 2, x.shape[2] // 2 * 2
        x = x[:, :H2, :W2]
        x = torch.cat((x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]), -1)
        x = self.reduction(self.norm(x))
        return _image(x) if image else x


class PatchExpand(nn.Module):
    def __init__(self, dim, scale, output_dim=None):
        super().__init__()
        self.scale, self.linear_first = scale, output_dim is None
        if output_dim is None:
            self.expand = nn.Linear(dim, scale * dim, bias=False)
            self.norm = nn.LayerNorm(dim // scale)
        else:
            self.expand = nn.Linear(dim // scale ** 2, output_dim, bias=False)
            self.norm = nn.LayerNorm(output_dim)

    def depth_to_space(self, x):
        B, H, W, C = x.shape
        s = self.scale
        return x.view(B, H, W, s, s, C // (s * s)).permute(0, 1, 3, 2, 4, 5).reshape(B, H * s, W * s, C // (s * s))

    def forward(self, x):          # image in, tokens out
        x = _tokens(x)
        x = self.depth_to_space(self.expand(x)) if self.linear_first else self.expand(self.depth_to_space(x))
        return self.norm(x)


class VSSMEncoder(nn.Module):
    """as MU builds it: one VSSBlock per level, `dims` equal on all levels, a REBNCONV stem (add_last), patch size 1, the
    downsample in front of the last level left out (skip_last_downsample)"""

    def __init__(self, in_ch, out_ch, dim, n_levels, drop_path_rate=0.2):
        super().__init__()
        self.rebnconvin = REBNCONV(in_ch, out_ch, 1)
        self.patch_embed = PatchEmbed2D(1, out_ch, dim)
        rates = torch.linspace(0, drop_path_rate, n_levels).tolist()
        self.layers = nn.ModuleList(VSSLayer(dim, 1, [rates[i]]) for i in range(n_levels))
        self.downsamples = nn.ModuleList(PatchMerging2D(dim, dim) for _ in range(n_levels - 2))

    def forward(self, x):
        x = self.rebnconvin(x)
        feats = [x]
        x = self.patch_embed(x)
        for i, layer in enumerate(self.layers):
            x = layer(x)
            feats.append(_image(x))
            if i < len(self.downsamples):
                x = self.downsamples[i](x)
        return feats


class VSSMDecoder(nn.Module):
    def __init__(self, out_ch, dim, n_levels, drop_path_rate=0.2):
        super().__init__()
        rates = torch.linspace(drop_path_rate, 0, (n_levels - 1) * 2).tolist()
        stages, expands, segs, fuse = [], [None], [], []   # level 0 meets its skip at the same resolution: no expand layer
        for s in range(1, n_levels):
            if s > 1:
                expands.append(PatchExpand(dim, 2, output_dim=dim))
            stages.append(VSSLayer(dim, 1, rates[s - 1:s]))
            segs.append(nn.Conv2d(dim, out_ch, 1))
            fuse.append(nn.Linear(2 * dim, dim))
        expands.append(PatchExpand(dim, 1))
        stages.append(nn.Identity())
        segs.append(nn.Conv2d(dim, out_ch, 1))
        self.stages = nn.ModuleList(stages)
        self.expand_layers = nn.ModuleList(expands)
        self.seg_layers = nn.ModuleList(segs)
        self.concat_back_dim = nn.ModuleList(fuse)

    def forward(self, feats):
        low = feats[-1]
        last = len(self.stages) - 1
        for s in range(last + 1):
            x = _tokens(low) if s == 0 else self.expand_layers[s](low)
            if s < last:
                x = self.concat_back_dim[s](torch.cat((x, _tokens(feats[-(s + 2)])), -1))
            low = _image(self.stages[s](x))
        return self.seg_layers[-1](low)


class MU(nn.Module):
    def __init__(self, in_ch, mid_ch, out_ch, n_levels):
        super().__init__()
        self.vssm_encoder = VSSMEncoder(in_ch, out_ch, mid_ch, n_levels)
        self.vssm_decoder = VSSMDecoder(out_ch, mid_ch, n_levels)

    def forward(self, x):
        feats = self.vssm_encoder(x)
        return self.vssm_decoder(feats) + feats[0]


class _X2Net(nn.Module):
    """macro wiring of m2net.py:883-956"""

    def fuse(self, k, up_tokens, skip):
        raise NotImplementedError

    def forward(self, x):
        h1 = self.stage1(x)
        h2 = self.stage2(self.patch_merging1(h1, image=True))
        h3 = self.stage3(self.patch_merging2(h2, image=True))
        h4 = self.stage4(self.patch_merging3(h3, image=True))
        h5 = self.stage5(self.patch_merging4(h4, image=True))
        h6 = self.stage6(self.pool56(h5))
        d5 = self.stage5d(torch.cat((upsample_like(h6, h5.shape[2:]), h5), 1))
        d4 = self.stage4d(self.fuse(4, self.patch_expand4d(d5), h4))
        d3 = self.stage3d(self.fuse(3, self.patch_expand3d(d4), h3))
        d2 = self.stage2d(self.fuse(2, self.patch_expand2d(d3), h2))
        d1 = self.stage1d(self.fuse(1, self.patch_expand1d(d2), h1))
        sides = [self.side1(d1), self.side2(d2), self.side3(d3), self.side4(d4), self.side5(d5), self.side6(h6)]
        full = sides[0].shape[2:]
        fused = self.outconv(torch.cat([sides[0]] + [upsample_like(s, full) for s in sides[1:]], 1))
        return (fused, *sides) if self.deep_supervision else fused


class M2Net(_X2Net):
    def __init__(self, in_ch, out_ch, deep_supervision=True):
        super().__init__()
        self.deep_supervision = deep_supervision
        self.stage1 = MU(in_ch, 16, 32, 7)
        self.patch_merging1 = PatchMerging2D(32)
        self.stage2 = MU(64, 32, 64, 6)
        self.patch_merging2 = PatchMerging2D(64)
        self.stage3 = MU(128, 64, 128, 5)
        self.patch_merging3 = PatchMerging2D(128)
        self.stage4 = MU(256, 128, 256, 4)
        self.patch_merging4 = PatchMerging2D(256)
        self.stage5 = RSU4F(512, 256, 512)
        self.pool56 = nn.MaxPool2d(2, stride=2, ceil_mode=True)
        self.stage6 = RSU4F(512, 256, 512)
        self.stage5d = RSU4F(1024, 256, 512)
        for k, c in ((4, 512), (3, 256), (2, 128), (1, 64)):
            setattr(self, f"patch_expand{k}d", PatchExpand(c, 2))
            setattr(self, f"concat_back_dim{k}d", nn.Linear(c, c // 2))
            setattr(self, f"stage{k}d", MU(c // 2, c // 4, c // 2, 8 - k))
        for i, c in enumerate((32, 64, 128, 256, 512, 512), 1):
            setattr(self, f"side{i}", nn.Conv2d(c, out_ch, 3, padding=1))
        self.outconv = nn.Conv2d(6 * out_ch, out_ch, 1)

    def fuse(self, k, up_tokens, skip):
        return _image(getattr(self, f"concat_back_dim{k}d")(torch.cat((up_tokens, _tokens(skip)), -1)))


class M2NetP(_X2Net):
    def __init__(self, in_ch, out_ch, deep_supervision=True):
        super().__init__()
        self.deep_supervision = deep_supervision
        self.stage1 = MU(in_ch, 16, 64, 7)
        self.patch_merging1 = PatchMerging2D(64, 64)
        for k, n in ((2, 6), (3, 5), (4, 4)):
            setattr(self, f"stage{k}", MU(64, 16, 64, n))
            setattr(self, f"patch_merging{k}", PatchMerging2D(64, 64))
        self.stage5 = RSU4F(64, 16, 64)
        self.pool56 = nn.MaxPool2d(2, stride=2, ceil_mode=True)
        self.stage6 = RSU4F(64, 16, 64)
        self.stage5d = RSU4F(128, 16, 128)
        for k in (4, 3, 2, 1):
            setattr(self, f"patch_expand{k}d", PatchExpand(128, 2))
            setattr(self, f"stage{k}d", MU(128, 16, 128, 8 - k))
        for i, c in enumerate((128, 128, 128, 128, 128, 64), 1):
            setattr(self, f"side{i}", nn.Conv2d(c, out_ch, 3, padding=1))
        self.outconv = nn.Conv2d(6 * out_ch, out_ch, 1)

    def fuse(self, k, up_tokens, skip):
        return torch.cat((_image(up_tokens), skip), 1)
