"""Mamba2 ("SSD") mixer - the block `mamba_ssm.modules.mamba2.Mamba2` that the reference's LightMamba2Net binds
(/root/reference/nnunetv2/nets/light_mamba2net.py:17 `from mamba_ssm.modules.mamba2 import Mamba2 as Mamba`, used by
MambaLayer :51-89 with d_state=16, d_conv=4, expand=2, headdim=get_nheaddim(...)).

mamba_ssm is a third-party dependency that is NOT vendored under /root/reference (pyproject.toml only pins torch for it),
so this file restates the published Mamba2 block (Dao & Gu 2024, mamba_ssm 2.2.x `Mamba2.__init__/forward`):

  zxbcdt = in_proj(u)                       -> [z (d_inner) | xBC (d_inner + 2*ngroups*d_state) | dt (nheads)]
  xBC    = silu(causal depthwise conv1d(xBC, width d_conv))
  x, B, C = split(xBC)
  per head h (headdim channels p):  dt_t = softplus(dt_t + dt_bias[h]);  a_t = exp(dt_t * A[h]),  A[h] = -exp(A_log[h])
      H_t = a_t * H_{t-1} + dt_t * x_t (p) (x) B_t (n);   y_t = H_t C_t + D[h] * x_t
  y = RMSNorm(y * silu(z)) * norm.weight    (gated RMS norm, norm_before_gate=False, one group)
  out = out_proj(y)

PARITY: pinned against an independent public implementation that IS available offline - HuggingFace transformers'
`Mamba2Mixer.torch_forward` (tests/golden/mamba2_mixer.npz, made by tools/make_mamba2_golden.py); the parameter names and
their order are those of mamba_ssm (in_proj, conv1d, dt_bias, A_log, D, norm.weight, out_proj).

MI355X mapping: the SSD recurrence is the selective scan with a per-head scalar decay, i.e. exactly the recurrence of
csrc/selective_scan.hip with A[(h, p), n] = A[h], delta[(h, p)] = dt[h], D[(h, p)] = D[h] and one B/C group - so the
block runs on the existing chunk-scan / causal-conv1d / gate kernels; the head -> channel broadcasts are views whose
gradients autograd sums back per head.  No CPU path: the kernels raise on CPU tensors.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

from ..mamba_block import causal_conv1d_fn, silu_gate
from ..selective_scan import selective_scan_fn
from ..token_linear import TokenLinear


class RMSNormGated(nn.Module):
    """mamba_ssm.ops.triton.layernorm_gated.RMSNorm(norm_before_gate=False, group_size=hidden): rms-normalises
    x * silu(z) over the channel axis.  `forward` takes the (B, C, L) layout the scan writes."""

    def __init__(self, hidden_size: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size))

    def forward(self, x_bcl: torch.Tensor, z_bcl: torch.Tensor) -> torch.Tensor:
        g = silu_gate(x_bcl, z_bcl)
        return g * torch.rsqrt(g.pow(2).mean(1, keepdim=True) + self.eps) * self.weight[None, :, None]


class Mamba2(nn.Module):
    def __init__(self, d_model, d_state=128, d_conv=4, conv_init=None, expand=2, headdim=64, ngroups=1,
                 A_init_range=(1, 16), dt_min=0.001, dt_max=0.1, dt_init_floor=1e-4, bias=False, conv_bias=True,
                 chunk_size=256, layer_idx=None):
        super().__init__()
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = self.d_ssm = expand * d_model
        self.headdim, self.ngroups, self.chunk_size, self.layer_idx = headdim, ngroups, chunk_size, layer_idx
        if d_state != 16 or ngroups != 1:
            raise NotImplementedError("nnuzoo_amd.Mamba2: d_state 16 / one B-C group (the LightMamba2Net configuration)")
        if self.d_ssm % headdim:
            raise ValueError("d_inner must be a multiple of headdim")
        self.nheads = self.d_ssm // headdim
        # registration and RNG order as mamba_ssm's, so that equal seeds give equal parameters
        self.in_proj = TokenLinear(d_model, 2 * self.d_inner + 2 * ngroups * d_state + self.nheads, bias=bias)
        conv_dim = self.d_ssm + 2 * ngroups * d_state
        self.conv1d = nn.Conv1d(conv_dim, conv_dim, bias=conv_bias, kernel_size=d_conv, groups=conv_dim, padding=d_conv - 1)
        if conv_init is not None:
            nn.init.uniform_(self.conv1d.weight, -conv_init, conv_init)
        self.act = nn.SiLU()
        dt = torch.exp(torch.rand(self.nheads) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)) \
            .clamp(min=dt_init_floor)
        self.dt_bias = nn.Parameter(dt + torch.log(-torch.expm1(-dt)))           # inverse softplus
        self.dt_bias._no_weight_decay = True
        self.A_log = nn.Parameter(torch.log(torch.empty(self.nheads, dtype=torch.float32).uniform_(*A_init_range)))
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.nheads))
        self.D._no_weight_decay = True
        self.norm = RMSNormGated(self.d_ssm, eps=1e-5)
        self.out_proj = TokenLinear(self.d_inner, d_model, bias=bias)

    def _per_channel(self, per_head: torch.Tensor) -> torch.Tensor:
        return per_head.float().repeat_interleave(self.headdim)

    def forward(self, u: torch.Tensor, seq_idx=None, inference_params=None) -> torch.Tensor:
        """u (B, L, d_model) -> (B, L, d_model)"""
        if inference_params is not None or seq_idx is not None:
            raise NotImplementedError("nnuzoo_amd.Mamba2: step-wise decoding / packed sequences are outside the hot path")
        if not u.is_cuda:
            raise RuntimeError("nnuzoo_amd.Mamba2 runs on MI355X through libnnuzoo_hip.so only (no CPU fallback)")
        Bt, L, _ = u.shape
        ds, n = self.d_ssm, self.d_state
        # in_proj produced directly in the channel-major (B, C, L) layout the conv / scan kernels read
        zxbcdt = (self.in_proj.weight @ u.reshape(Bt * L, -1).t()).view(-1, Bt, L).transpose(0, 1)
        if self.in_proj.bias is not None:
            zxbcdt = zxbcdt + self.in_proj.bias[None, :, None]
        z, xBC, dt = zxbcdt[:, :ds], zxbcdt[:, ds:2 * ds + 2 * n], zxbcdt[:, 2 * ds + 2 * n:]
        xBC = causal_conv1d_fn(xBC.contiguous(), self.conv1d.weight, self.conv1d.bias, activation="silu")
        x, Bm, Cm = xBC[:, :ds], xBC[:, ds:ds + n], xBC[:, ds + n:]
        delta = dt.repeat_interleave(self.headdim, dim=1)                       # (B, d_ssm, L): dt of the channel's head
        A = self._per_channel(-torch.exp(self.A_log.float()))[:, None].expand(ds, n)
        y = selective_scan_fn(x, delta, A, Bm, Cm, self._per_channel(self.D), z=None,
                              delta_bias=self._per_channel(self.dt_bias), delta_softplus=True)
        y = self.norm(y, z.contiguous())
        return F.linear(y.transpose(1, 2), self.out_proj.weight, self.out_proj.bias)

    def step(self, *a, **k):
        raise NotImplementedError("nnuzoo_amd.Mamba2: step-wise decoding is outside the segmentation hot path")

    allocate_inference_cache = step
