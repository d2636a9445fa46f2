"""M2Net ("SS2D^2Net") and its small variant M2NetP for MI355X - same public classes, constructor signatures,
sub-module / parameter names and shapes as /root/reference/nnunetv2/nets/m2net.py, so checkpoints interchange:

  SS2D              :39-225     2-D selective-scan block (4 scan directions, d_state 16, expand 2)
  VSSBlock/VSSLayer :513-595    x + DropPath(SS2D(LayerNorm(x)))
  PatchEmbed2D      :486-510
  VSSMEncoder       :598-710,   VSSMDecoder :359-483,   MU :713-765   (inner U-net of one U^2 stage)
  M2Net             :805-971,   M2NetP :1011-1184
  get_m2net_from_plans / get_m2netp_from_plans :1187-1232

The SS2D block body runs on hand-written gfx950 kernels (nnuzoo_amd/ss2d_scan.py): depthwise conv + SiLU + scan layouts
(csrc/ss2d_dwconv.hip), the chunk scan in cross-scan mode - four directions by index arithmetic, delta and A formed in
the kernel, fp32 as the reference forces it (`.float()` at :185-191) - (csrc/selective_scan.hip), direction merge
(csrc/ss2d_layout.hip) and the gated output norm (csrc/layer_norm.hip); every LayerNorm is nnuzoo_amd.layer_norm.LayerNorm
and the tall token Linears are nnuzoo_amd.token_linear.TokenLinear.  `SS2D.fused_cross_scan = False` selects the
reference's op-by-op formulation around `selective_scan_fn` (parity tests).  The remaining Linear / RSU4F conv layers are
library ops.  There is no eager fallback for the scan: CPU tensors raise.
"""
from __future__ import annotations

import math
from typing import List, Tuple, Union

import torch

from .. import backends as _backends
import torch.nn.functional as F
from torch import nn

from ..token_linear import TokenLinear
from ..layer_norm import LayerNorm, layer_norm_gate, layer_norm_skip

from .. import ss2d_scan
from ..selective_scan import selective_scan_fn
from ..utilities.network_initialization import InitWeights_He
from .common2d import DropPath, PatchExpand, PatchMerging2D, REBNCONV, RSU4F, _upsample_like, residual_drop_path


class SS2D(nn.Module):
    K = 4  # scan directions: row-major, column-major and their reversals
    fused_cross_scan = True  # False: the reference's op-by-op formulation around selective_scan_fn (parity tests)
    fused_dwconv = True      # False: library depthwise conv + SiLU in front of the fused cross-scan

    def __init__(self, d_model, d_state=16, d_conv=3, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, dropout=0., conv_bias=True, bias=False,
                 device=None, dtype=None):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        K, R, N, Di = self.K, self.dt_rank, d_state, self.d_inner

        # creation order = the reference's (m2net.py:70-110): the RNG stream of a seeded construction is then the same
        # (named_parameters() lists a module's bare parameters before its children whatever the order of creation)
        self.in_proj = TokenLinear(d_model, Di * 2, bias=bias, **fk)
        self.conv2d = nn.Conv2d(Di, Di, groups=Di, bias=conv_bias, kernel_size=d_conv, padding=(d_conv - 1) // 2, **fk)
        self.act = nn.SiLU()
        xp = [nn.Linear(Di, R + 2 * N, bias=False, **fk).weight for _ in range(K)]
        self.x_proj_weight = nn.Parameter(torch.stack(xp, dim=0))                      # (K, R + 2N, Di)
        dts = [self.dt_init(R, Di, dt_scale, dt_init, dt_min, dt_max, dt_init_floor, **fk) for _ in range(K)]
        self.dt_projs_weight = nn.Parameter(torch.stack([t.weight for t in dts], dim=0))  # (K, Di, R)
        self.dt_projs_bias = nn.Parameter(torch.stack([t.bias for t in dts], dim=0))      # (K, Di)
        self.A_logs = self.A_log_init(N, Di, copies=K, merge=True)                     # (K * Di, N)
        self.Ds = self.D_init(Di, copies=K, merge=True)                                # (K * Di)

        self.selective_scan = selective_scan_fn
        self.out_norm = LayerNorm(Di)
        self.out_proj = TokenLinear(Di, d_model, bias=bias, **fk)
        self.dropout = nn.Dropout(dropout) if dropout > 0. else None

    # ---- initialisers (same distributions as the reference, :113-168) --------------------------------------------
    @staticmethod
    def dt_init(dt_rank, d_inner, dt_scale=1.0, dt_init="random", dt_min=0.001, dt_max=0.1, dt_init_floor=1e-4,
                **fk):
        proj = nn.Linear(dt_rank, d_inner, bias=True, **fk)
        std = dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(proj.weight, std)
        elif dt_init == "random":
            nn.init.uniform_(proj.weight, -std, std)
        else:
            raise NotImplementedError
        dt = torch.exp(torch.rand(d_inner, **fk) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
        dt = dt.clamp(min=dt_init_floor)
        with torch.no_grad():
            proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))  # inverse softplus
        proj.bias._no_reinit = True
        return proj

    @staticmethod
    def A_log_init(d_state, d_inner, copies=1, device=None, merge=True):
        A = torch.arange(1, d_state + 1, dtype=torch.float32, device=device).repeat(d_inner, 1)
        A_log = torch.log(A)
        if copies > 1:
            A_log = A_log.unsqueeze(0).repeat(copies, 1, 1)
            if merge:
                A_log = A_log.flatten(0, 1)
        p = nn.Parameter(A_log)
        p._no_weight_decay = True
        return p

    @staticmethod
    def D_init(d_inner, copies=1, device=None, merge=True):
        D = torch.ones(d_inner, device=device)
        if copies > 1:
            D = D.unsqueeze(0).repeat(copies, 1)
            if merge:
                D = D.flatten(0, 1)
        p = nn.Parameter(D)
        p._no_weight_decay = True
        return p

    # ---- forward ------------------------------------------------------------------------------------------------
    def forward_core(self, x: torch.Tensor):
        """x: (B, Di, H, W) -> four (B, Di, L) outputs, all in row-major token order."""
        B, Di, H, W = x.shape
        L, K, N, R = H * W, self.K, self.d_state, self.dt_rank
        rows = x.reshape(B, Di, L)
        cols = x.transpose(2, 3).reshape(B, Di, L)
        fwd = torch.stack([rows, cols], dim=1)                       # (B, 2, Di, L)
        xs = torch.cat([fwd, fwd.flip(-1)], dim=1)                   # (B, 4, Di, L)
        proj = torch.einsum("bkdl,kcd->bkcl", xs, self.x_proj_weight)
        dts, Bs, Cs = torch.split(proj, [R, N, N], dim=2)
        dts = torch.einsum("bkrl,kdr->bkdl", dts, self.dt_projs_weight)
        y = self.selective_scan(
            xs.float().reshape(B, K * Di, L), dts.contiguous().float().reshape(B, K * Di, L),
            -torch.exp(self.A_logs.float()).view(-1, N), Bs.float().contiguous(), Cs.float().contiguous(),
            self.Ds.float().view(-1), z=None, delta_bias=self.dt_projs_bias.float().view(-1), delta_softplus=True,
            return_last_state=False).view(B, K, Di, L)
        assert y.dtype == torch.float32
        back = y[:, 2:4].flip(-1)

        def untranspose(t):  # column-major token order -> row-major
            return t.reshape(B, Di, W, H).transpose(2, 3).reshape(B, Di, L)

        return y[:, 0], back[:, 0], untranspose(y[:, 1]), untranspose(back[:, 1])

    def forward(self, x: torch.Tensor, **kwargs):
        B, H, W, C = x.shape
        x, z = self.in_proj(x).chunk(2, dim=-1)
        fused = self.fused_cross_scan and x.is_cuda and self.d_state == 16 and 1 <= self.dt_rank <= 8 \
            and self.d_inner % 4 == 0 and B * self.d_inner <= 65535 and x.dtype in (torch.float16, torch.float32)
        if fused and self.fused_dwconv and ss2d_scan.dwconv_supported(self.conv2d):
            _backends.note(self, "hip")
            # conv + SiLU + both scan layouts in one kernel, directions by index arithmetic inside the scan kernels,
            # gated output norm in one kernel (nnuzoo_amd/ss2d_scan.py, layer_norm.py)
            y = ss2d_scan.ss2d_conv_cross_scan(x, self.conv2d, self.x_proj_weight, self.dt_projs_weight,
                                               self.dt_projs_bias, self.A_logs, self.Ds)
            y = layer_norm_gate(y, z, self.out_norm.weight, self.out_norm.bias, self.out_norm.eps,
                                feeds_linear=True)                      # LN(y) * silu(z) -> out_proj
        else:
            _backends.note(self, "hip-opbyop" if x.is_cuda else "library",
                           why="op-by-op formulation around selective_scan_fn (fused_cross_scan off or unsupported shape)")
            x = self.act(self.conv2d(x.permute(0, 3, 1, 2).contiguous()))
            if fused:
                y = ss2d_scan.ss2d_cross_scan(x, self.x_proj_weight, self.dt_projs_weight, self.dt_projs_bias,
                                              self.A_logs, self.Ds)
                y = layer_norm_gate(y, z, self.out_norm.weight, self.out_norm.bias, self.out_norm.eps,
                                    feeds_linear=True)
            else:
                y1, y2, y3, y4 = self.forward_core(x)
                y = (y1 + y2 + y3 + y4).transpose(1, 2).reshape(B, H, W, -1)
                y = self.out_norm(y) * F.silu(z)
        out = self.out_proj(y)
        return self.dropout(out) if self.dropout is not None else out


class VSSBlock(nn.Module):
    def __init__(self, hidden_dim: int = 0, drop_path: float = 0, norm_layer=LayerNorm, attn_drop_rate: float = 0,
                 d_state: int = 16, **kwargs):
        super().__init__()
        self.ln_1 = norm_layer(hidden_dim)
        if isinstance(self.ln_1, LayerNorm):
            self.ln_1.feeds_linear = True     # its only consumer is SS2D.in_proj
        self.self_attention = SS2D(d_model=hidden_dim, dropout=attn_drop_rate, d_state=d_state, **kwargs)
        self.drop_path = DropPath(drop_path)

    def forward(self, input: torch.Tensor):
        # (ln_1(input), input): the residual stream is taken from the norm's second output so that both gradients of `input` meet
        # inside the LayerNorm backward kernel instead of in an add launch of the autograd engine (80 per M2Net step)
        y, skip = layer_norm_skip(self.ln_1, input)
        return residual_drop_path(skip, self.self_attention(y), self.drop_path)


class VSSLayer(nn.Module):
    def __init__(self, dim, depth, attn_drop=0., drop_path=0., norm_layer=LayerNorm, downsample=None,
                 use_checkpoint=False, d_state=16):
        super().__init__()
        self.dim, self.use_checkpoint = dim, use_checkpoint
        self.blocks = nn.ModuleList([
            VSSBlock(hidden_dim=dim, drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                     norm_layer=norm_layer, attn_drop_rate=attn_drop, d_state=d_state) for i in range(depth)])
        # the reference runs a no-op re-init here that only advances the RNG (m2net.py:571-578):
        # kaiming_uniform_ on a detached CLONE of every `out_proj.weight`, visited in nn.Module.apply order.  Replayed so
        # that a seeded construction draws the same stream as the reference's (bitwise equal parameters, see
        # tests/test_state_dict_manifest.py::test_seeded_construction_checksums)
        def _advance_rng(module: nn.Module):
            for name, p in module.named_parameters():
                if name in ["out_proj.weight"]:
                    nn.init.kaiming_uniform_(p.clone().detach_(), a=math.sqrt(5))

        self.apply(_advance_rng)
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return self.downsample(x) if self.downsample is not None else x


class PatchEmbed2D(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None, **kwargs):
        super().__init__()
        if isinstance(patch_size, int):
            patch_size = (patch_size, patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.proj._nnz_fp32_master = patch_size == (1, 1)     # read as fp32 by the token kernels: no fp16 shadow (param_shadow.py)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        from .. import sepconv32
        xt = x.permute(0, 2, 3, 1)
        if xt.is_contiguous() and sepconv32.pointwise_ok(self.proj, xt):
            # patch size 1 (every MU stage of the X^2-Nets): a Linear over the channels of token-major activations - the fp32 MFMA
            # Linear kernels on the rows as they lie (fp16 rows in / out under autocast), no layout copy, no library call
            _backends.note(self.proj, "hip-f32")
            x = sepconv32.pointwise_tokens(self.proj, xt)
        else:
            x = self.proj(x).permute(0, 2, 3, 1)
        return self.norm(x) if self.norm is not None else x


def _vssm_init(m: nn.Module):
    """trunc_normal(.02) Linear weights, zero Linear biases, unit LayerNorm (m2net.py:666-682)."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


class VSSMEncoder(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, depths=[2, 2, 9, 2], dims=[96, 192, 384, 768], d_state=16,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, norm_layer=LayerNorm, patch_norm=True,
                 use_checkpoint=False, skip_first_downsample: bool = False, skip_last_downsample: bool = False,
                 add_last: bool = False, out_ch: int = None):
        super().__init__()
        self.num_layers = len(depths)
        self.add_last, self.skip_last_downsample, self.skip_first_downsample = add_last, skip_last_downsample, \
            skip_first_downsample
        if isinstance(dims, int):
            dims = [int(dims * 2 ** i) for i in range(self.num_layers)]
        if add_last:
            self.rebnconvin = REBNCONV(in_chans, out_ch, dirate=1)
        self.embed_dim, self.dims = dims[0], dims
        self.patch_embed = PatchEmbed2D(patch_size=patch_size, in_chans=out_ch if add_last else in_chans,
                                        embed_dim=self.embed_dim, norm_layer=norm_layer if patch_norm else None)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        self.downsamples = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(VSSLayer(dim=dims[i], depth=depths[i],
                                        d_state=math.ceil(dims[0] / 6) if d_state is None else d_state,
                                        attn_drop=attn_drop_rate, drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])],
                                        norm_layer=norm_layer, downsample=None, use_checkpoint=use_checkpoint))
            if i < self.num_layers - 1:
                if (i == 0 and skip_first_downsample) or (i == self.num_layers - 2 and skip_last_downsample):
                    continue
                self.downsamples.append(PatchMerging2D(input_dim=dims[i], scale=2, output_features=dims[i + 1],
                                                       norm_layer=norm_layer))
        self.apply(_vssm_init)

    def forward(self, x):
        feats = []
        if self.add_last:
            x = self.rebnconvin(x)
            feats.append(x)
        else:
            feats.append(None)
        x = self.pos_drop(self.patch_embed(x))
        for s, layer in enumerate(self.layers):
            x = layer(x)
            feats.append(x.permute(0, 3, 1, 2))
            if s < len(self.downsamples):
                if s == 0 and self.skip_first_downsample:
                    continue
                x = self.downsamples[s](x)
        return feats


class VSSMDecoder(nn.Module):
    def __init__(self, num_classes: int, deep_supervision, features_per_stage: Union[Tuple[int, ...], List[int]] = None,
                 depths: Union[Tuple[int, ...], List[int]] = None, drop_path_rate: float = 0.2, d_state: int = 16,
                 skip_first_expand: bool = False, patch_size: int = 4):
        super().__init__()
        self.skip_first_expand, self.deep_supervision, self.num_classes = skip_first_expand, deep_supervision, num_classes
        chans = features_per_stage
        n_enc = len(chans)
        dpr = [x.item() for x in torch.linspace(drop_path_rate, 0, (n_enc - 1) * 2)]
        depths = depths or [2] * n_enc
        stages, expands, segs, fuse = [], [], [], []
        skip = 0
        for s in range(1, n_enc):
            below, skip = chans[-s], chans[-(s + 1)]
            if s == 1 and skip_first_expand:
                expands.append(None)
            else:
                expands.append(PatchExpand(dim=below, scale=2, output_dim=below, norm_layer=LayerNorm))
            stages.append(VSSLayer(dim=skip, depth=1, attn_drop=0., drop_path=dpr[sum(depths[:s - 1]):sum(depths[:s])],
                                   d_state=math.ceil(2 * skip / 6) if d_state is None else d_state,
                                   norm_layer=LayerNorm, downsample=None, use_checkpoint=False))
            segs.append(nn.Conv2d(skip, num_classes, 1, 1, 0, bias=True))
            segs[-1]._nnz_fp32_master = True
            fuse.append(TokenLinear(2 * skip, skip))      # nn.Linear subclass: MFMA token kernels under fp16 autocast
        expands.append(PatchExpand(dim=chans[0], scale=patch_size, norm_layer=LayerNorm))
        if isinstance(expands[-1].norm, LayerNorm):
            expands[-1].norm.feeds_linear = True     # its only consumer is the 1x1 output convolution: fp16 rows straight from the kernel
        stages.append(nn.Identity())
        segs.append(nn.Conv2d(skip, num_classes, 1, 1, 0, bias=True))
        segs[-1]._nnz_fp32_master = True
        self.stages = nn.ModuleList(stages)
        self.expand_layers = nn.ModuleList(expands)
        self.seg_layers = nn.ModuleList(segs)
        self.concat_back_dim = nn.ModuleList(fuse)

    def forward(self, skips):
        low = skips[-1]
        outs = []
        last = len(self.stages) - 1
        for s in range(len(self.stages)):
            if s == 0 and self.skip_first_expand:
                x = low.permute(0, 2, 3, 1)
            else:
                x = self.expand_layers[s](low)
            if s < last:
                x = self.concat_back_dim[s](torch.cat((x, skips[-(s + 2)].permute(0, 2, 3, 1)), -1))
            xt = self.stages[s](x)
            x = xt.permute(0, 3, 1, 2)
            if self.deep_supervision:
                outs.append(self._seg(self.seg_layers[s], xt, x))
            elif s == last:
                outs.append(self._seg(self.seg_layers[-1], xt, x))
            low = x
        outs = outs[::-1]
        return outs if self.deep_supervision else outs[0]

    @staticmethod
    def _seg(conv, tokens, nchw):
        """the 1x1 output convolution of a stage, on the stage's token-major rows when the token kernels take them"""
        from .. import sepconv32
        if tokens.is_contiguous() and sepconv32.pointwise_ok(conv, tokens):
            _backends.note(conv, "hip-f32")
            return sepconv32.pointwise_tokens(conv, tokens).permute(0, 3, 1, 2)
        return conv(nchw)


class MU(nn.Module):
    """inner Mamba U-net of one U^2 stage"""

    def __init__(self, in_ch: int, mid_ch, out_ch: int, n_layers: int, skip_last_downsample: bool = False,
                 patch_size: int = 4, add_last: bool = False):
        super().__init__()
        self.add_last = add_last
        feats, depths = [mid_ch] * n_layers, [1] * n_layers
        self.vssm_encoder = VSSMEncoder(in_chans=in_ch, patch_size=patch_size, depths=depths, dims=feats,
                                        skip_first_downsample=False, skip_last_downsample=skip_last_downsample,
                                        add_last=add_last, out_ch=out_ch if add_last else None, drop_path_rate=0.2)
        self.vssm_decoder = VSSMDecoder(num_classes=out_ch, deep_supervision=False, features_per_stage=feats,
                                        drop_path_rate=0.2, d_state=16, depths=depths,
                                        skip_first_expand=skip_last_downsample, patch_size=patch_size)

    def forward(self, x):
        skips = self.vssm_encoder(x)
        out = self.vssm_decoder(skips)
        return out + skips[0] if self.add_last else out

    @torch.no_grad()
    def freeze_encoder(self):
        for name, p in self.vssm_encoder.named_parameters():
            if "patch_embed" not in name:
                p.requires_grad = False

    @torch.no_grad()
    def unfreeze_encoder(self):
        for p in self.vssm_encoder.parameters():
            p.requires_grad = True


class _U2Forward:
    """Macro wiring shared by the X^2-Nets (m2net.py:883-956): 4 encoder stages + 2 dilated RSU4F stages, 4 decoder
    stages, six side outputs fused by a 1x1 conv.  `_fuse(k, up, skip)` differs between the variants."""

    def _fuse(self, k: int, up_tokens, skip_nchw):
        raise NotImplementedError

    def forward(self, x):
        from ..droppath_draws import DrawTable
        with DrawTable(self, x.shape[0], x.device):      # every stochastic-depth draw of the pass from one launch
            return self._forward(x)

    def _forward(self, x):
        h1 = self.stage1(x)
        h2 = self.stage2(self.patch_merging1(h1, permute=True))
        h3 = self.stage3(self.patch_merging2(h2, permute=True))
        h4 = self.stage4(self.patch_merging3(h3, permute=True))
        h5 = self.stage5(self.patch_merging4(h4, permute=True))
        h6 = self.stage6(self.pool56(h5))
        h5d = self.stage5d(torch.cat((_upsample_like(h6, h5.shape[2:]), h5), 1))
        h4d = self.stage4d(self._fuse(4, self.patch_expand4d(h5d), h4))
        h3d = self.stage3d(self._fuse(3, self.patch_expand3d(h4d), h3))
        h2d = self.stage2d(self._fuse(2, self.patch_expand2d(h3d), h2))
        h1d = self.stage1d(self._fuse(1, self.patch_expand1d(h2d), h1))
        d1, d2, d3 = self._head(self.side1, h1d), self._head(self.side2, h2d), self._head(self.side3, h3d)
        d4, d5, d6 = self._head(self.side4, h4d), self._head(self.side5, h5d), self._head(self.side6, h6)
        full = d1.shape[2:]
        d0 = self._head(self.outconv, torch.cat([d1] + [_upsample_like(d, full) for d in (d2, d3, d4, d5, d6)], 1))
        return (d0, d1, d2, d3, d4, d5, d6) if self.deep_supervision else d0

    @staticmethod
    def _head(mod, x):
        """a side / fuse convolution: 1x1 convolutions to <= 8 channels of an fp32 device step (SwT2Net's) run on csrc/sepconv32.hip
        head1x1_* - straight off the token-major stage output, no layout copy, no library call; everything else is the module"""
        from .. import rebnconv, sepconv32
        conv = getattr(mod, "conv", mod)
        if sepconv32.head1x1_ok(conv, x):
            _backends.note(mod, "hip-f32")
            return sepconv32.head1x1(conv, x)
        if rebnconv.head3x3_ok(conv, x):
            # 3x3 side head of an fp16-autocast step (m2net.py:874-880): one 32-channel block of the tap-table conv kernels
            _backends.note(mod, "hip")
            return rebnconv.head3x3(conv, x)
        return mod(x)

    def _mark_heads(self):
        """the side / fuse convolutions are read as fp32 master weights by the HIP head kernels: no fp16 shadow (param_shadow.py)"""
        for name in ("side1", "side2", "side3", "side4", "side5", "side6", "outconv"):
            getattr(self, name)._nnz_fp32_master = True

    def _encoder_groups(self):
        return [self.stage1, self.stage2, self.stage3, self.stage4, self.stage5, self.stage6, self.patch_merging1,
                self.patch_merging2, self.patch_merging3, self.patch_merging4, self.pool56]

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


class M2Net(_U2Forward, nn.Module):
    def __init__(self, in_ch: int, out_ch: int, deep_supervision: bool):
        nn.Module.__init__(self)
        self.deep_supervision = deep_supervision

        def mu(i, m, o, n):
            return MU(in_ch=i, mid_ch=m, out_ch=o, n_layers=n, skip_last_downsample=True, patch_size=1, add_last=True)

        self.stage1 = mu(in_ch, 16, 32, 7)
        self.patch_merging1 = PatchMerging2D(32, scale=2)
        self.stage2 = mu(64, 32, 64, 6)
        self.patch_merging2 = PatchMerging2D(64, scale=2)
        self.stage3 = mu(128, 64, 128, 5)
        self.patch_merging3 = PatchMerging2D(128, scale=2)
        self.stage4 = mu(256, 128, 256, 4)
        self.patch_merging4 = PatchMerging2D(256, scale=2)
        self.stage5 = RSU4F(512, 256, 512)
        self.pool56 = nn.MaxPool2d(2, stride=2, ceil_mode=True)
        self.stage6 = RSU4F(512, 256, 512)
        self.stage5d = RSU4F(1024, 256, 512)
        self.patch_expand4d = PatchExpand(dim=512, scale=2, norm_layer=LayerNorm)
        self.concat_back_dim4d = TokenLinear(512, 256)
        self.stage4d = mu(256, 128, 256, 4)
        self.patch_expand3d = PatchExpand(dim=256, scale=2, norm_layer=LayerNorm)
        self.concat_back_dim3d = TokenLinear(256, 128)
        self.stage3d = mu(128, 64, 128, 5)
        self.patch_expand2d = PatchExpand(dim=128, scale=2, norm_layer=LayerNorm)
        self.concat_back_dim2d = TokenLinear(128, 64)
        self.stage2d = mu(64, 32, 64, 6)
        self.patch_expand1d = PatchExpand(dim=64, scale=2, norm_layer=LayerNorm)
        self.concat_back_dim1d = TokenLinear(64, 32)
        self.stage1d = mu(32, 16, 32, 7)
        for i, c in enumerate([32, 64, 128, 256, 512, 512], 1):
            setattr(self, f"side{i}", nn.Conv2d(c, out_ch, 3, padding=1))
        self.outconv = nn.Conv2d(6 * out_ch, out_ch, 1)
        self._mark_heads()

    def _fuse(self, k, up_tokens, skip_nchw):
        lin = getattr(self, f"concat_back_dim{k}d")
        return lin(torch.cat((up_tokens, skip_nchw.permute(0, 2, 3, 1)), -1)).permute(0, 3, 1, 2)


class M2NetP(_U2Forward, nn.Module):
    def __init__(self, in_ch: int, out_ch: int, deep_supervision: bool, spatial_dims: int = 2):
        nn.Module.__init__(self)
        self.deep_supervision = deep_supervision

        def mu(i, o, n):
            return MU(in_ch=i, mid_ch=16, out_ch=o, n_layers=n, skip_last_downsample=True, patch_size=1, add_last=True)

        self.stage1 = mu(in_ch, 64, 7)
        self.patch_merging1 = PatchMerging2D(64, scale=2, output_features=64)
        self.stage2 = mu(64, 64, 6)
        self.patch_merging2 = PatchMerging2D(64, scale=2, output_features=64)
        self.stage3 = mu(64, 64, 5)
        self.patch_merging3 = PatchMerging2D(64, scale=2, output_features=64)
        self.stage4 = mu(64, 64, 4)
        self.patch_merging4 = PatchMerging2D(64, scale=2, output_features=64)
        self.stage5 = RSU4F(64, 16, 64)
        self.pool56 = nn.MaxPool2d(2, stride=2, ceil_mode=True)
        self.stage6 = RSU4F(64, 16, 64)
        self.stage5d = RSU4F(128, 16, 128)
        self.patch_expand4d = PatchExpand(dim=128, scale=2, norm_layer=LayerNorm)
        self.stage4d = mu(128, 128, 4)
        self.patch_expand3d = PatchExpand(dim=128, scale=2, norm_layer=LayerNorm)
        self.stage3d = mu(128, 128, 5)
        self.patch_expand2d = PatchExpand(dim=128, scale=2, norm_layer=LayerNorm)
        self.stage2d = mu(128, 128, 6)
        self.patch_expand1d = PatchExpand(dim=128, scale=2, norm_layer=LayerNorm)
        self.stage1d = mu(128, 128, 7)
        for i, c in enumerate([128, 128, 128, 128, 128, 64], 1):
            setattr(self, f"side{i}", nn.Conv2d(c, out_ch, 3, padding=1))
        self.outconv = nn.Conv2d(6 * out_ch, out_ch, 1)
        self._mark_heads()

    def _fuse(self, k, up_tokens, skip_nchw):
        return torch.cat([up_tokens.permute(0, 3, 1, 2), skip_nchw], 1)


def _heads(plans_manager, dataset_json) -> int:
    if plans_manager is not None and hasattr(plans_manager, "get_label_manager"):
        return plans_manager.get_label_manager(dataset_json).num_segmentation_heads
    return len(dataset_json["labels"])


def get_m2net_from_plans(plans_manager, dataset_json: dict, configuration_manager, num_input_channels: int,
                         deep_supervision: bool = True, use_pretrain: bool = True):
    model = M2Net(in_ch=num_input_channels, out_ch=_heads(plans_manager, dataset_json),
                  deep_supervision=deep_supervision)
    model.apply(InitWeights_He(1e-2))
    return model


def get_m2netp_from_plans(plans_manager, dataset_json: dict, configuration_manager, num_input_channels: int,
                          deep_supervision: bool = True, use_pretrain: bool = True):
    model = M2NetP(in_ch=num_input_channels, out_ch=_heads(plans_manager, dataset_json),
                   deep_supervision=deep_supervision)
    model.apply(InitWeights_He(1e-2))
    return model
