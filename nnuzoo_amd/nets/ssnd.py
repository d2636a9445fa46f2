"""SSND - the N-D (2-D: 4 directions, 3-D: 6 directions) selective-scan block of SSND2Net, same constructor,
parameter names and shapes as /root/reference/nnunetv2/nets/ssnd2net.py:73-318, on the gfx950 scan kernel.

Reproduced reference behaviour, including quirk 3 of SURVEY.md §8b: in the 3-D branch BOTH the `wzh` and the `hwz`
output terms are built from direction 1's scan output viewed as (W, Z, H) (ssnd2net.py:291-298); direction 2 is
scanned and discarded.  Parity means reproducing that, so we do (tests/golden/ssnd3d.npz pins it).

The rest of SSND2Net (GSC gate, N-D patch merge/expand, outer U^2 wiring, ssnd2net.py:321-1775) is not built in this
round; this module is the operator-level piece that differs from m2net.SS2D.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

from ..token_linear import TokenLinear
from ..layer_norm import LayerNorm

from .. import ss2d_scan
from ..layer_norm import layer_norm_gate
from ..selective_scan import selective_scan_fn
from .common2d import Convolution
from .m2net import SS2D


class SSND(nn.Module):
    def __init__(self, spatial_dims: int, factorization_type: str, d_model: int, d_state=16, d_conv=3, expand=2,
                 dt_rank="auto", dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0, dt_init_floor=1e-4,
                 dropout=0., conv_bias=True, bias=False, device=None, dtype=None, dilation=1):
        super().__init__()
        if factorization_type != "cross-scan" or spatial_dims not in (2, 3):
            raise Exception("Factorization and spatial_dims are not supported!")
        fk = {"device": device, "dtype": dtype}
        self.spatial_dims, self.factorization_type = spatial_dims, factorization_type
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.k = K = 2 * spatial_dims
        R, N, Di = self.dt_rank, d_state, self.d_inner
        # creation order = the reference's (ssnd2net.py:108-160): same RNG stream for a seeded construction
        self.in_proj = TokenLinear(d_model, Di * 2, bias=bias, **fk)
        self.convnd = Convolution(spatial_dims, Di, Di, groups=Di, bias=conv_bias, kernel_size=d_conv,
                                  padding=(d_conv - 1) // 2 * dilation if dilation != 1 else (d_conv - 1) // 2,
                                  conv_only=True, dilation=dilation)
        self.act = nn.SiLU()
        xp = [nn.Linear(Di, R + 2 * N, bias=False, **fk).weight for _ in range(K)]
        self.x_proj_weight = nn.Parameter(torch.stack(xp, dim=0))
        dts = [SS2D.dt_init(R, Di, dt_scale, dt_init, dt_min, dt_max, dt_init_floor, **fk) for _ in range(K)]
        self.dt_projs_weight = nn.Parameter(torch.stack([t.weight for t in dts], dim=0))
        self.dt_projs_bias = nn.Parameter(torch.stack([t.bias for t in dts], dim=0))
        self.A_logs = SS2D.A_log_init(N, Di, copies=K, merge=True)
        self.Ds = SS2D.D_init(Di, copies=K, merge=True)
        self.selective_scan = selective_scan_fn
        self.out_norm = LayerNorm(Di)
        self.out_proj = TokenLinear(Di, d_model, bias=bias, **fk)
        self.dropout = nn.Dropout(dropout) if dropout > 0. else None

    def forward_core(self, x: torch.Tensor):
        """x: (B, Di, [Z,] H, W) -> (B, [Z,] H, W, Di)"""
        B, Di = x.shape[:2]
        sp = x.shape[2:]
        L, K, N, R = int(torch.tensor(sp).prod()), self.k, self.d_state, self.dt_rank
        if self.spatial_dims == 2:
            H, W = sp
            fwd = torch.stack([x.reshape(B, Di, L), x.transpose(2, 3).reshape(B, Di, L)], dim=1)
        else:
            Z, H, W = sp
            fwd = torch.stack([x.reshape(B, Di, L), x.permute(0, 1, 4, 2, 3).reshape(B, Di, L),   # w z h
                               x.permute(0, 1, 3, 4, 2).reshape(B, Di, L)], dim=1)                 # h w z
        xs = torch.cat([fwd, fwd.flip(-1)], dim=1)
        proj = torch.einsum("bkdl,kcd->bkcl", xs, self.x_proj_weight)
        dts, Bs, Cs = torch.split(proj, [R, N, N], dim=2)
        dts = torch.einsum("bkrl,kdr->bkdl", dts, self.dt_projs_weight)
        y = self.selective_scan(
            xs.float().reshape(B, K * Di, L), dts.contiguous().float().reshape(B, K * Di, L),
            -torch.exp(self.A_logs.float()).view(-1, N), Bs.float().contiguous(), Cs.float().contiguous(),
            self.Ds.float().view(-1), z=None, delta_bias=self.dt_projs_bias.float().view(-1), delta_softplus=True,
            return_last_state=False).view(B, K, Di, L)
        back = y[:, K // 2:].flip(-1)
        if self.spatial_dims == 2:
            def unt(t):
                return t.reshape(B, Di, W, H).transpose(2, 3).reshape(B, Di, L)
            out = y[:, 0] + back[:, 0] + unt(y[:, 1]) + unt(back[:, 1])
            return out.transpose(1, 2).reshape(B, H, W, Di)
        d1, b1 = y[:, 1].reshape(B, Di, W, Z, H), back[:, 1].reshape(B, Di, W, Z, H)
        # reference: "b c w z h -> b c z h w" and (on the SAME direction-1 tensors) "b c h w z -> b c z h w"
        out = (y[:, 0] + back[:, 0]
               + d1.permute(0, 1, 3, 4, 2).reshape(B, Di, L) + b1.permute(0, 1, 3, 4, 2).reshape(B, Di, L)
               + d1.permute(0, 1, 4, 2, 3).reshape(B, Di, L) + b1.permute(0, 1, 4, 2, 3).reshape(B, Di, L))
        return out.transpose(1, 2).reshape(B, Z, H, W, Di)

    def forward(self, x: torch.Tensor):
        x, z = self.in_proj(x).chunk(2, dim=-1)
        if self.spatial_dims == 2 and SS2D.fused_cross_scan and x.is_cuda and self.d_state == 16 \
                and 1 <= self.dt_rank <= 8 and self.d_inner % 4 == 0 and x.shape[0] * self.d_inner <= 65535 \
                and x.dtype in (torch.float16, torch.float32):
            # the 2-D block is SS2D's computation (same 4 directions, same order): fused kernels of nnuzoo_amd/ss2d_scan.py
            conv = self.convnd.conv
            args = (self.x_proj_weight, self.dt_projs_weight, self.dt_projs_bias, self.A_logs, self.Ds)
            if SS2D.fused_dwconv and ss2d_scan.dwconv_supported(conv):
                y = ss2d_scan.ss2d_conv_cross_scan(x, conv, *args)
            else:                                                   # dilated depthwise conv: library conv in front
                y = ss2d_scan.ss2d_cross_scan(self.act(self.convnd(x.permute(0, 3, 1, 2).contiguous())), *args)
            y = layer_norm_gate(y, z, self.out_norm.weight, self.out_norm.bias, self.out_norm.eps,
                                feeds_linear=True)
            out = self.out_proj(y)
            return self.dropout(out) if self.dropout is not None else out
        perm = (0, 3, 1, 2) if self.spatial_dims == 2 else (0, 4, 1, 2, 3)
        x = self.act(self.convnd(x.permute(*perm).contiguous()))
        y = self.out_norm(self.forward_core(x)) * F.silu(z)
        out = self.out_proj(y)
        return self.dropout(out) if self.dropout is not None else out
