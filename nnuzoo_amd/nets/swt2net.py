"""SwT2Net (Swin-transformer U^2-Net) for MI355X - same public classes, constructor signatures, sub-module /
parameter / buffer names and shapes as /root/reference/nnunetv2/nets/swt2net.py, so checkpoints interchange:

  PatchEmbedding :412-432, PatchMerging :435-464, PatchExpanding :467-478, FinalPatchExpanding :481-493, Mlp :496-515
  WindowAttention :518-619 (incl. the int64 buffer `relative_position_index`)
  SwinTransformerBlock :622-661 (pads TOP/LEFT to a multiple of 7 and crops from the end; shift mask -100)
  BasicBlock / BasicBlockUp :664-740,  SwinTransformerUnet :743-869,  SwT2Net :909-1156
  get_swt2net_from_plans :1372-1393

The attention core (roll, window partition, per-head softmax(q k^T + bias + mask) v, merge, un-partition, roll back)
is ONE hand-written gfx950 MFMA kernel (nnuzoo_amd.window_attention, csrc/window_attention.hip) in fp32, as the
reference's Swin trainers run without autocast; in the fp32 device step a whole SwinTransformerBlock is one autograd node of
five forward and seven backward launches (nnuzoo_amd/swin_block.py: pad / crop, both LayerNorms, DropPath and the residual adds
live in the prologues / epilogues of the fp32 MFMA Linear kernels, csrc/dense32.hip).
"""
from __future__ import annotations

from functools import partial
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from ..layer_norm import LayerNorm, layer_norm_skip
from ..token_linear import TokenLinear, mlp_gelu

from ..utilities.network_initialization import InitWeights_He
from ..swin_block import fused_block_ok, swin_block_forward
from ..window_attention import window_attention_core
from .common2d import Convolution, PatchExpand, PatchMerging2D, _ResidualDropPathFn, get_dwconv_layer
from .common2d import RSU4F as _RSU4F
from .m2net import _U2Forward, _heads


class REBNCONV(nn.Module):
    """swt2net.py:17-31: depthwise 3x3 + pointwise 1x1 (both bias-free, `dirate` accepted but unused) -> BN -> ReLU"""

    def __init__(self, in_ch=3, out_ch=3, dirate=1):
        super().__init__()
        self.conv_s1 = get_dwconv_layer(spatial_dims=2, in_channels=in_ch, out_channels=out_ch)
        self.bn_s1 = nn.BatchNorm2d(out_ch)
        self.relu_s1 = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.relu_s1(self.bn_s1(self.conv_s1(x)))


class RSU4F(_RSU4F):
    block = REBNCONV


class DropPath(nn.Module):
    """per-sample stochastic depth, floor(keep + U[0,1)) form of swt2net.py:395-409"""

    def __init__(self, drop_prob: float = 0.):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = (keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)).floor_()
        return x.div(keep) * mask


class _PadCropFn(torch.autograd.Function):
    """pad=True: zero rows / columns in FRONT of a (B, H, W, C) fp32 map (F.pad(x, (0, 0, px, 0, py, 0))); pad=False: the
    crop x[:, py:, px:, :] as a contiguous tensor.  Each is the other's backward, one launch either way (csrc/residual.hip
    pad_crop_kernel) - ATen runs the pad and the crop's backward as a fill plus a copy."""

    @staticmethod
    def forward(ctx, x, py, px, pad):
        from .._lib import call, ptr, stream_ptr
        B, H, W, C = x.shape
        ctx.cfg = (py, px, pad)
        if pad:
            out = torch.empty((B, H + py, W + px, C), dtype=torch.float32, device=x.device)
            call("nnz_pad_top_left", ptr(x), ptr(out), B, H, W, C, py, px, stream_ptr())
        else:
            out = torch.empty((B, H - py, W - px, C), dtype=torch.float32, device=x.device)
            call("nnz_crop_top_left", ptr(x), ptr(out), B, H - py, W - px, C, py, px, stream_ptr())
        return out

    @staticmethod
    def backward(ctx, g):
        py, px, pad = ctx.cfg
        return _PadCropFn.apply(g.contiguous(), py, px, not pad), None, None, None


def _pad_crop_ok(x, rows):
    return x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] % 4 == 0 \
        and x.shape[0] * rows <= 65535


class PatchEmbedding(nn.Module):
    def __init__(self, patch_size: int = 4, in_c: int = 3, embed_dim: int = 96, norm_layer=None):
        super().__init__()
        self.patch_size = patch_size
        self.proj = nn.Conv2d(in_c, embed_dim, kernel_size=(patch_size,) * 2, stride=(patch_size,) * 2)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        _, _, H, W = x.shape
        p = self.patch_size
        if H % p or W % p:
            x = F.pad(x, (0, p - W % p, 0, p - H % p, 0, 0))  # right/bottom, full extra patch when one side divides
        # the non-overlapping conv (kernel = stride = p) as space-to-depth + one GEMM on the conv's own parameters: same
        # arithmetic, lands token-major directly, and avoids the library's strided-conv weight gradient (23 ms per call
        # at 512^2 in fp32: `profiles/r01_swt2net_step_kernels.txt`)
        B, C, H, W = x.shape
        x = x.view(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B, H // p, W // p, C * p * p)
        from .. import backends as _backends
        from .. import sepconv32
        if sepconv32.patch_embed_ok(self.proj, x):
            _backends.note(self.proj, "hip-f32")       # fp32 MFMA Linear kernels (csrc/dense32.hip), weight gradient in the grouped launch
            return self.norm(sepconv32.pointwise_tokens(self.proj, x))
        if x.is_cuda:
            _backends.note(self.proj, "library", why="patch embedding outside the fp32 device step / K not a multiple of 4")
        return self.norm(F.linear(x, self.proj.weight.view(self.proj.out_channels, -1), self.proj.bias))


class PatchMerging(nn.Module):
    def __init__(self, dim: int, norm_layer=LayerNorm):
        super().__init__()
        self.dim = dim
        self.norm = norm_layer(4 * dim)
        self.reduction = TokenLinear(4 * dim, 2 * dim, bias=False)

    def forward(self, x):
        _, H, W, _ = x.shape
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        # torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1) of the reference (swt2net.py:452-456)
        # as one permuted copy: channel block k = 2 * (column parity) + (row parity).  Same values; the backward is one
        # strided copy instead of four zero fills, four slice copies and three adds
        B, H2, W2, C = x.shape[0], x.shape[1] // 2, x.shape[2] // 2, x.shape[3]
        x = x.view(B, H2, 2, W2, 2, C).permute(0, 1, 3, 4, 2, 5).reshape(B, H2, W2, 4 * C) if x.is_contiguous() else \
            torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
        return self.reduction(self.norm(x))


def _depth_to_space(x, p):
    B, H, W, C = x.shape
    c = C // (p * p)
    return x.view(B, H, W, p, p, c).permute(0, 1, 3, 2, 4, 5).reshape(B, H * p, W * p, c)


class PatchExpanding(nn.Module):
    def __init__(self, dim: int, norm_layer=LayerNorm):
        super().__init__()
        self.dim = dim
        self.expand = TokenLinear(dim, 2 * dim, bias=False)
        self.norm = norm_layer(dim // 2)

    def forward(self, x):
        return self.norm(_depth_to_space(self.expand(x), 2))


class FinalPatchExpanding(nn.Module):
    def __init__(self, dim: int, norm_layer=LayerNorm, patch_size: int = 4):
        super().__init__()
        self.dim = dim
        self.expand = TokenLinear(dim, (patch_size ** 2) * dim, bias=False)
        self.norm = norm_layer(dim)
        self.patch_size = patch_size

    def forward(self, x):
        return self.norm(_depth_to_space(self.expand(x), self.patch_size))


class Mlp(nn.Module):
    def __init__(self, in_features: int, hidden_features: int = None, out_features: int = None, act_layer=nn.GELU,
                 drop: float = 0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = TokenLinear(in_features, hidden_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = TokenLinear(hidden_features, out_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        if isinstance(self.act, nn.GELU) and self.act.approximate == "none" and self.drop1.p == 0:
            return self.drop2(mlp_gelu(x, self.fc1, self.fc2))     # fp32 step: one fused chain on csrc/dense32.hip
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class WindowAttention(nn.Module):
    def __init__(self, dim: int, window_size: int, num_heads: int, qkv_bias: Optional[bool] = True,
                 attn_drop: Optional[float] = 0., proj_drop: Optional[float] = 0., shift: bool = False):
        super().__init__()
        if window_size != 7:
            raise NotImplementedError("the HIP window-attention kernel is specialised to the zoo's 7x7 windows")
        if attn_drop:
            raise NotImplementedError("attention dropout is 0 everywhere in the zoo and not implemented in the kernel")
        self.window_size, self.num_heads = window_size, num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.shift_size = window_size // 2 if shift else 0
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        ar = torch.arange(window_size)
        coords = torch.stack(torch.meshgrid([ar, ar], indexing="ij")).flatten(1)       # (2, 49)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0) + (window_size - 1)
        self.register_buffer("relative_position_index", rel[:, :, 0] * (2 * window_size - 1) + rel[:, :, 1])
        self.qkv = TokenLinear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = TokenLinear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.softmax = nn.Softmax(dim=-1)

    def _index_i32(self) -> torch.Tensor:
        idx = getattr(self, "_idx32", None)
        if idx is None or idx.device != self.relative_position_index.device:
            idx = self.relative_position_index.to(torch.int32).contiguous()
            self._idx32 = idx  # plain attribute (not a buffer): the state_dict keeps only the reference's int64 buffer
        return idx

    def forward(self, x):
        """x: (B, H, W, C) with H, W multiples of the window size (the block pads)."""
        qkv = self.qkv(x)  # per-token Linear: commutes with the roll / window partition the kernel folds in
        out = window_attention_core(qkv, self.relative_position_bias_table, self._index_i32(), self.num_heads,
                                    self.shift_size, self.scale)
        return self.proj_drop(self.proj(out))


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=7, shift=False, mlp_ratio=4., qkv_bias=True, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=LayerNorm):
        super().__init__()
        self.window_size = window_size
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size=window_size, num_heads=num_heads, qkv_bias=qkv_bias,
                                    attn_drop=attn_drop, proj_drop=drop, shift=shift)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def _residual(self, inp, y):
        """inp + drop_path(y) (swt2net.py:379-388) as ONE kernel each way (csrc/residual.hip) instead of div, mul and add;
        the per-sample mask is drawn exactly like the reference's DropPath (torch.rand of shape (B, 1, 1, 1), floor)"""
        dp = self.drop_path
        ok = y.is_cuda and inp.shape == y.shape and y.dtype == torch.float32 and inp.dtype == torch.float32 \
            and y.is_contiguous() and inp.is_contiguous() and (y.numel() // y.shape[0]) % 4 == 0
        if not ok:
            return inp + dp(y)
        if isinstance(dp, DropPath) and dp.drop_prob > 0. and dp.training:
            keep = 1 - dp.drop_prob
            from ..droppath_draws import uniform
            draws = uniform(y.shape[0], y.device)
            if keep > 0.0:   # floor(keep + draws) is made inside the residual kernel (fp32 add, then floor, like torch's)
                return _ResidualDropPathFn.apply(inp, y, draws, 1.0 / keep, keep)
            return _ResidualDropPathFn.apply(inp, y, (keep + draws).floor_(), 1.0)
        return _ResidualDropPathFn.apply(inp, y, None, 1.0)

    def forward(self, x):
        if self.window_size == 7 and fused_block_ok(self, x):
            return swin_block_forward(self, x)      # five launches forward, seven backward (nnuzoo_amd/swin_block.py)
        _, H, W, _ = x.shape
        ws = self.window_size
        pad = H % ws != 0 or W % ws != 0
        py, px = ws - H % ws, ws - W % ws
        fused = pad and _pad_crop_ok(x, H + py)
        if pad:  # top/left padding, a full extra window on an axis that already divides (reference quirk, :643-645)
            x = _PadCropFn.apply(x, py, px, True) if fused else F.pad(x, (0, 0, px, 0, py, 0))
        n, x = layer_norm_skip(self.norm1, x)       # (norm(x), x): both gradients of x meet inside the norm's backward
        x = self._residual(x, self.attn(n))
        n, x = layer_norm_skip(self.norm2, x)
        x = self._residual(x, self.mlp(n))
        if not pad:
            return x
        return _PadCropFn.apply(x, py, px, False) if fused and x.is_contiguous() else x[:, -H:, -W:, :]


def _stage_drop_path(depths, drop_path, index):
    dpr = [r.item() for r in torch.linspace(0, drop_path, sum(depths))]
    return dpr[sum(depths[:index]):sum(depths[:index + 1])]


def _swin_blocks(dim, depth, num_head, window_size, mlp_ratio, qkv_bias, drop_rate, attn_drop_rate, rates, norm_layer):
    return nn.ModuleList([
        SwinTransformerBlock(dim=dim, num_heads=num_head, window_size=window_size, shift=bool(i % 2),
                             mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop=drop_rate, attn_drop=attn_drop_rate,
                             drop_path=rates[i], norm_layer=norm_layer) for i in range(depth)])


class BasicBlock(nn.Module):
    def __init__(self, index: int, embed_dim: int = 96, window_size: int = 7, depths: tuple = (2, 2, 6, 2),
                 num_heads: tuple = (3, 6, 12, 24), mlp_ratio: float = 4., qkv_bias: bool = True, drop_rate: float = 0.,
                 attn_drop_rate: float = 0., drop_path: float = 0.1, norm_layer=LayerNorm,
                 patch_merging: bool = True):
        super().__init__()
        dim = embed_dim * 2 ** index
        self.blocks = _swin_blocks(dim, depths[index], num_heads[index], window_size, mlp_ratio, qkv_bias, drop_rate,
                                   attn_drop_rate, _stage_drop_path(depths, drop_path, index), norm_layer)
        self.downsample = PatchMerging(dim=dim, norm_layer=norm_layer) if patch_merging else None

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return self.downsample(x) if self.downsample is not None else x


class BasicBlockUp(nn.Module):
    def __init__(self, index: int, embed_dim: int = 96, window_size: int = 7, depths: tuple = (2, 2, 6, 2),
                 num_heads: tuple = (3, 6, 12, 24), mlp_ratio: float = 4., qkv_bias: bool = True, drop_rate: float = 0.,
                 attn_drop_rate: float = 0., drop_path: float = 0.1, patch_expanding: bool = True,
                 norm_layer=LayerNorm):
        super().__init__()
        index = len(depths) - index - 2
        dim = embed_dim * 2 ** index
        self.blocks = _swin_blocks(dim, depths[index], num_heads[index], window_size, mlp_ratio, qkv_bias, drop_rate,
                                   attn_drop_rate, _stage_drop_path(depths, drop_path, index), norm_layer)
        self.upsample = PatchExpanding(dim=dim, norm_layer=norm_layer) if patch_expanding else nn.Identity()

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return self.upsample(x)


class SwinTransformerUnet(nn.Module):
    def __init__(self, patch_size: int = 4, in_ch: int = 3, out_ch: int = 1000, embed_dim: int = 96,
                 window_size: int = 7, depths: tuple = (2, 2, 6, 2), num_heads: tuple = (3, 6, 12, 24),
                 mlp_ratio: float = 4., qkv_bias: bool = True, drop_rate: float = 0., attn_drop_rate: float = 0.,
                 drop_path_rate: float = 0.1, norm_layer=LayerNorm, patch_norm: bool = True, add_last: bool = False):
        super().__init__()
        self.add_last, self.window_size, self.depths, self.num_heads = add_last, window_size, depths, num_heads
        self.num_layers, self.embed_dim = len(depths), embed_dim
        common = dict(depths=depths, embed_dim=embed_dim, num_heads=num_heads, drop_path=drop_path_rate,
                      window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop_rate=drop_rate,
                      attn_drop_rate=attn_drop_rate, norm_layer=norm_layer)
        if add_last:
            self.rebnconvin = get_dwconv_layer(2, in_ch, out_ch)
        self.patch_embed = PatchEmbedding(patch_size=patch_size, in_c=in_ch, embed_dim=embed_dim,
                                          norm_layer=norm_layer if patch_norm else None)
        self.pos_drop = nn.Dropout(p=drop_rate)
        n = self.num_layers
        self.layers = nn.ModuleList([BasicBlock(index=i, patch_merging=i != n - 1, **common) for i in range(n)])
        self.first_patch_expanding = PatchExpanding(dim=embed_dim * 2 ** (n - 1), norm_layer=norm_layer)
        self.layers_up = nn.ModuleList([BasicBlockUp(index=i, patch_expanding=i < n - 2, **common)
                                        for i in range(n - 1)])
        self.skip_connection_layers = nn.ModuleList([
            TokenLinear(embed_dim * 2 ** (n - 2 - i) * 2, embed_dim * 2 ** (n - 2 - i)) for i in range(n - 1)])
        self.norm_up = norm_layer(embed_dim)
        self.final_patch_expanding = FinalPatchExpanding(dim=embed_dim, norm_layer=norm_layer, patch_size=patch_size)
        self.head = nn.Conv2d(embed_dim, out_ch, kernel_size=(1, 1), bias=False)
        self.apply(self.init_weights)

    @staticmethod
    def init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, x):
        from .. import backends as _backends
        from .. import sepconv32
        res = None
        if self.add_last:
            if sepconv32.stem_ok(self.rebnconvin, x):
                # depthwise 3x3 + pointwise 1x1 of the stage's residual branch, token-major on csrc/sepconv32.hip / dense32.hip
                _backends.note(self.rebnconvin, "hip-f32")
                res = sepconv32.stem_forward(self.rebnconvin, x)
            else:
                if x.is_cuda:
                    _backends.note(self.rebnconvin, "library", why="stem with a channel count that is not a multiple of 4 (the "
                                                                   "1-channel network input) / autocast / non-fp32")
                res = self.rebnconvin(x)
        x = self.pos_drop(self.patch_embed(x))
        saved = []
        for layer in self.layers:
            saved.append(x)
            x = layer(x)
        x = self.first_patch_expanding(x)
        for i, layer in enumerate(self.layers_up):
            skip = saved[len(saved) - i - 2]
            x = x[:, :skip.shape[1], :skip.shape[2], :]  # drop the rows/cols that came from odd-size padding
            x = layer(self.skip_connection_layers[i](torch.cat([x, skip], -1)))
        x = self.final_patch_expanding(self.norm_up(x))
        if sepconv32.pointwise_ok(self.head, x):
            _backends.note(self.head, "hip-f32")          # the 1x1 head IS a token Linear (fp32 MFMA, csrc/dense32.hip)
            x = sepconv32.pointwise_tokens(self.head, x).permute(0, 3, 1, 2)
        else:
            x = self.head(x.permute(0, 3, 1, 2))
        return x + res if self.add_last else x


class SwT2Net(_U2Forward, nn.Module):
    def __init__(self, in_ch: int, out_ch: int, deep_supervision: bool):
        nn.Module.__init__(self)
        self.spatial_dims = 2
        self.deep_supervision = deep_supervision
        ln = partial(LayerNorm, eps=1e-6)

        def su(patch, i, o, embed, heads):
            return SwinTransformerUnet(patch_size=patch, in_ch=i, out_ch=o, depths=(2, 2, 4, 2), embed_dim=embed,
                                       num_heads=heads, window_size=7, qkv_bias=True, mlp_ratio=4, drop_path_rate=0.1,
                                       drop_rate=0, attn_drop_rate=0, norm_layer=ln, add_last=True)

        self.stage1 = su(4, in_ch, 32, 32, (2, 2, 4, 8))
        self.patch_merging1 = PatchMerging2D(32, scale=2)
        self.stage2 = su(4, 64, 64, 64, (2, 4, 8, 16))
        self.patch_merging2 = PatchMerging2D(64, scale=2)
        self.stage3 = su(2, 128, 128, 96, (3, 6, 12, 24))
        self.patch_merging3 = PatchMerging2D(128, scale=2)
        self.stage4 = su(1, 256, 256, 96, (3, 6, 12, 24))
        self.patch_merging4 = PatchMerging2D(256, scale=2)
        self.stage5 = RSU4F(512, 256, 512)
        self.pool56 = nn.MaxPool2d(2, stride=2, ceil_mode=True)
        self.stage6 = RSU4F(512, 256, 512)
        self.stage5d = RSU4F(1024, 256, 512)
        self.patch_expand4d = PatchExpand(dim=512, scale=2, norm_layer=LayerNorm)
        self.concat_back_dim4d = TokenLinear(512, 256)
        self.stage4d = su(1, 256, 256, 96, (3, 6, 12, 24))
        self.patch_expand3d = PatchExpand(dim=256, scale=2, norm_layer=LayerNorm)
        self.concat_back_dim3d = TokenLinear(256, 128)
        self.stage3d = su(2, 128, 128, 96, (3, 6, 12, 24))
        self.patch_expand2d = PatchExpand(dim=128, scale=2, norm_layer=LayerNorm)
        self.concat_back_dim2d = TokenLinear(128, 64)
        self.stage2d = su(4, 64, 64, 64, (2, 4, 8, 16))
        self.patch_expand1d = PatchExpand(dim=64, scale=2, norm_layer=LayerNorm)
        self.concat_back_dim1d = TokenLinear(64, 32)
        self.stage1d = su(4, 32, 32, 32, (2, 2, 4, 8))
        for i, c in enumerate([32, 64, 128, 256, 512, 512], 1):
            setattr(self, f"side{i}", Convolution(2, c, out_ch, kernel_size=1, padding=0, conv_only=True))
        self.outconv = Convolution(2, 6 * out_ch, out_ch, kernel_size=1, conv_only=True)

    def _fuse(self, k, up_tokens, skip_nchw):
        lin = getattr(self, f"concat_back_dim{k}d")
        return lin(torch.cat((up_tokens, skip_nchw.permute(0, 2, 3, 1)), -1)).permute(0, 3, 1, 2)


def get_swt2net_from_plans(plans_manager, dataset_json: dict, configuration_manager, num_input_channels: int,
                           deep_supervision: bool = True, use_pretrain: bool = True):
    model = SwT2Net(in_ch=num_input_channels, out_ch=_heads(plans_manager, dataset_json),
                    deep_supervision=deep_supervision)
    model.apply(InitWeights_He(1e-2))
    return model
