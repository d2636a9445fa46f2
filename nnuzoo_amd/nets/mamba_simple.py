"""`Mamba` (1-D selective-state-space block, optional bidirectional / slice-direction variants) for MI355X.

Same constructor, parameters, initialisation sequence and state_dict as the reference's vendored module
/root/reference/nnunetv2/nets/seg_mamba/mamba_simple.py:37-190 (used by SegMamba, segmamba.py:74-81; the `mamba_ssm.Mamba`
the reference binds in lm2net.py:14,70 and mamba_nd2net.py:26 has the bimamba_type="none" subset of it).  forward
(:190-357) runs on the HIP operators of nnuzoo_amd.mamba_block (causal conv1d + SiLU, selective scan, z gate); the
projections are library GEMMs.  `use_fast_path` is accepted and ignored: there is one path.  Step-wise decoding
(`inference_params`, `step`, `allocate_inference_cache`) is not part of the segmentation hot path and raises.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..mamba_block import mamba_inner_fn_no_out_proj


def _s4d_real_log(d_inner, d_state, device):
    return torch.log(torch.arange(1, d_state + 1, dtype=torch.float32, device=device).repeat(d_inner, 1).contiguous())


class Mamba(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False, use_fast_path=True,
                 layer_idx=None, device=None, dtype=None, bimamba_type="none", nslices=5, _single_direction_params=False):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(self.expand * self.d_model)
        self.dt_rank = math.ceil(self.d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path, self.layer_idx = use_fast_path, layer_idx
        self.bimamba_type, self.nslices = bimamba_type, nslices
        if d_state != 16:
            raise NotImplementedError("nnuzoo_amd.Mamba: d_state must be 16 (the scan kernel's state width)")

        def conv():
            return nn.Conv1d(self.d_inner, self.d_inner, bias=conv_bias, kernel_size=d_conv, groups=self.d_inner,
                             padding=d_conv - 1, **fk)

        # registration and RNG order follow the reference line by line, so equal seeds give equal parameters
        self.in_proj = nn.Linear(self.d_model, self.d_inner * 2, bias=bias, **fk)
        self.conv1d = conv()
        self.activation = "silu"
        self.act = nn.SiLU()
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **fk)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **fk)
        dt_init_std = self.dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(self.dt_proj.weight, dt_init_std)
        elif dt_init == "random":
            nn.init.uniform_(self.dt_proj.weight, -dt_init_std, dt_init_std)
        else:
            raise NotImplementedError
        dt = torch.exp(torch.rand(self.d_inner, **fk) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)) \
            .clamp(min=dt_init_floor)
        inv_dt = dt + torch.log(-torch.expm1(-dt))  # inverse softplus
        with torch.no_grad():
            self.dt_proj.bias.copy_(inv_dt)
        self.dt_proj.bias._no_reinit = True
        self.A_log = nn.Parameter(_s4d_real_log(self.d_inner, self.d_state, device))
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D._no_weight_decay = True
        if _single_direction_params:
            # parameter set of `mamba_ssm.Mamba` (class MambaSSM below): no backward / slice direction tensors
            if bimamba_type != "none":
                raise ValueError("the single-direction parameter set has no bimamba variants")
            self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias, **fk)
            return
        # backward direction
        self.A_b_log = nn.Parameter(_s4d_real_log(self.d_inner, self.d_state, device))
        self.A_b_log._no_weight_decay = True
        self.conv1d_b = conv()
        self.x_proj_b = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **fk)
        self.dt_proj_b = nn.Linear(self.dt_rank, self.d_inner, bias=True, **fk)
        self.D_b = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D_b._no_weight_decay = True
        # slice ("spatial") direction
        self.A_s_log = nn.Parameter(_s4d_real_log(self.d_inner, self.d_state, device))
        self.A_s_log._no_weight_decay = True
        self.conv1d_s = conv()
        self.x_proj_s = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **fk)
        self.dt_proj_s = nn.Linear(self.dt_rank, self.d_inner, bias=True, **fk)
        self.D_s = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D_s._no_weight_decay = True
        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias, **fk)

    def _branch(self, xz, conv, x_proj, dt_proj, A_log, D):
        return mamba_inner_fn_no_out_proj(xz, conv.weight, conv.bias, x_proj.weight, dt_proj.weight,
                                          -torch.exp(A_log.float()), None, None, D.float(),
                                          delta_bias=dt_proj.bias.float(), delta_softplus=True)

    def forward(self, hidden_states, inference_params=None):
        """hidden_states (B, L, D) -> (B, L, D)"""
        if inference_params is not None:
            raise NotImplementedError("nnuzoo_amd.Mamba: step-wise decoding is outside the segmentation hot path")
        if not hidden_states.is_cuda:
            raise RuntimeError("nnuzoo_amd.Mamba runs on MI355X through libnnuzoo_hip.so only (no CPU fallback)")
        batch, seqlen, _ = hidden_states.shape
        # (B, L, D) x W_in^T, produced directly in the (B, 2*d_inner, L) layout the conv / scan read
        xz = (self.in_proj.weight @ hidden_states.reshape(batch * seqlen, -1).t()).view(-1, batch, seqlen).transpose(0, 1)
        if self.in_proj.bias is not None:
            xz = xz + self.in_proj.bias.to(xz.dtype)[None, :, None]
        out = self._branch(xz, self.conv1d, self.x_proj, self.dt_proj, self.A_log, self.D)
        if self.bimamba_type in ("v2", "v3"):
            out_b = self._branch(xz.flip([-1]), self.conv1d_b, self.x_proj_b, self.dt_proj_b, self.A_b_log, self.D_b)
            out = out + out_b.flip([-1])
        if self.bimamba_type == "v3":
            # slice direction: tokens regrouped so that the sequence walks across the nslices chunks first
            ns = self.nslices
            xz_s = torch.stack(xz.chunk(ns, dim=-1), dim=-1).flatten(-2)
            out_s = self._branch(xz_s, self.conv1d_s, self.x_proj_s, self.dt_proj_s, self.A_s_log, self.D_s)
            out = out + out_s.reshape(batch, self.d_inner, seqlen // ns, ns).permute(0, 1, 3, 2).flatten(-2)
        return F.linear(out.transpose(1, 2), self.out_proj.weight, self.out_proj.bias)

    def step(self, *a, **k):
        raise NotImplementedError("nnuzoo_amd.Mamba: step-wise decoding is outside the segmentation hot path")

    allocate_inference_cache = step


class MambaSSM(Mamba):
    """`mamba_ssm.Mamba` as the reference binds it (`from mamba_ssm import Mamba`: nets/mamba_nd2net.py:26, lm2net.py:14,
    LightMUNet.py:6): the same block with the one-direction parameter set - state_dict keys in_proj.weight, conv1d.weight,
    conv1d.bias, x_proj.weight, dt_proj.weight, dt_proj.bias, A_log, D, out_proj.weight (same construction / RNG order as
    the vendored class above minus its `_b` / `_s` tensors).  mamba_ssm is not installed here and its version is not
    pinned by the reference: key names restated from the published 1.x / 2.x module."""

    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False, use_fast_path=True,
                 layer_idx=None, device=None, dtype=None):
        super().__init__(d_model, d_state, d_conv, expand, dt_rank, dt_min, dt_max, dt_init, dt_scale, dt_init_floor,
                         conv_bias, bias, use_fast_path, layer_idx, device, dtype, bimamba_type="none",
                         _single_direction_params=True)
