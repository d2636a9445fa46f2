"""TEST INFRASTRUCTURE ONLY - CPU restatement (plain torch, fp32) of the reference's SSND block, the N-D selective-scan block of
SSND2Net (/root/reference/nnunetv2/nets/ssnd2net.py:73-318; forward_core :239-302, forward :304-318) with `cross-scan`
factorisation: 4 scan orders in 2-D (rows, columns, their reversals - the computation of m2net.SS2D), 6 in 3-D (z h w, w z h,
h w z and their reversals).  Reference quirk kept (:291-298): in 3-D BOTH the `w z h` and the `h w z` output
terms are read from scan order 1 viewed as (W, Z, H); order 2 is scanned and not used.  Only tests/ import this file.

PINNED by the reference's own module: tests/test_oracle_ssnd.py on tests/golden/ssnd2d.npz and ssnd3d.npz (output, dx and every
parameter gradient of the reference's SSND, parameters by golden_util.det_fill)."""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .selective_scan import selective_scan_torch


class SSND(nn.Module):
    def __init__(self, spatial_dims, d_model, d_state=16, expand=2):
        super().__init__()
        assert spatial_dims in (2, 3)
        self.nd, self.K = spatial_dims, 2 * spatial_dims
        Di, N, R, K = expand * d_model, d_state, math.ceil(d_model / 16), 2 * spatial_dims
        self.Di, self.N, self.R = Di, N, R
        self.in_proj = nn.Linear(d_model, 2 * Di, bias=False)
        self.convnd = nn.Sequential()
        self.convnd.add_module("conv", (nn.Conv2d if spatial_dims == 2 else nn.Conv3d)(Di, Di, 3, padding=1, groups=Di))
        self.x_proj_weight = nn.Parameter(torch.randn(K, R + 2 * N, Di) * Di ** -0.5)
        self.dt_projs_weight = nn.Parameter((torch.rand(K, Di, R) * 2 - 1) * R ** -0.5)
        self.dt_projs_bias = nn.Parameter(torch.full((K, Di), 0.01))
        self.A_logs = nn.Parameter(torch.log(torch.arange(1, N + 1, dtype=torch.float32)).repeat(K * Di, 1))
        self.Ds = nn.Parameter(torch.ones(K * Di))
        self.out_norm = nn.LayerNorm(Di)
        self.out_proj = nn.Linear(Di, d_model, bias=False)

    def scans(self, x):
        """x (B, Di, *spatial) -> (B, *spatial, Di): the sum of the scan orders' outputs, each brought back to the layout of x"""
        B, Di = x.shape[:2]
        sp = tuple(x.shape[2:])
        L, K, N, R = math.prod(sp), self.K, self.N, self.R
        if self.nd == 2:
            orders = [x.flatten(2), x.transpose(2, 3).flatten(2)]
        else:
            orders = [x.flatten(2), x.permute(0, 1, 4, 2, 3).flatten(2), x.permute(0, 1, 3, 4, 2).flatten(2)]
        seq = torch.stack(orders + [o.flip(-1) for o in orders], 1)                       # (B, K, Di, L)
        proj = torch.einsum("bkdl,kcd->bkcl", seq, self.x_proj_weight)
        dt, Bm, Cm = proj.split((R, N, N), 2)
        dt = torch.einsum("bkrl,kdr->bkdl", dt, self.dt_projs_weight)
        y = selective_scan_torch(seq.reshape(B, K * Di, L), dt.reshape(B, K * Di, L), -torch.exp(self.A_logs.float()),
                                 Bm, Cm, self.Ds.float(), self.dt_projs_bias.reshape(-1).float(), True).view(B, K, Di, L)
        fwd, bwd = y[:, :K // 2], y[:, K // 2:].flip(-1)
        if self.nd == 2:
            H, W = sp

            def cols_to_rows(t):
                return t.reshape(B, Di, W, H).transpose(2, 3).reshape(B, Di, L)

            out = fwd[:, 0] + bwd[:, 0] + cols_to_rows(fwd[:, 1]) + cols_to_rows(bwd[:, 1])
        else:
            Z, H, W = sp
            a, b = fwd[:, 1].reshape(B, Di, W, Z, H), bwd[:, 1].reshape(B, Di, W, Z, H)
            out = fwd[:, 0] + bwd[:, 0]
            for t in (a, b):          # order 1 read as (w z h) AND as (h w z): the reference's own indexing, order 2 unused
                out = out + t.permute(0, 1, 3, 4, 2).reshape(B, Di, L) + t.permute(0, 1, 4, 2, 3).reshape(B, Di, L)
        return out.transpose(1, 2).reshape(B, *sp, Di)

    def forward(self, x):          # (B, *spatial, C)
        x, z = self.in_proj(x).chunk(2, -1)
        to_channels_first = (0, self.nd + 1) + tuple(range(1, self.nd + 1))
        x = F.silu(self.convnd(x.permute(*to_channels_first)))
        return self.out_proj(self.out_norm(self.scans(x)) * F.silu(z))
