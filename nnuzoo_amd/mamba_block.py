"""The 1-D Mamba block's operators around the selective scan, on the HIP kernels of csrc/mamba_block.hip:

  causal_conv1d_fn(x, weight, bias, activation="silu")   -- causal_conv1d's function as bound by
        /root/reference/nnunetv2/nets/seg_mamba/mamba_simple.py:13-16 and selective_scan_interface.py:652
  silu_gate(y, z) = y * silu(z)                          -- the `z` branch of selective_scan_fn
        (selective_scan_ref, selective_scan_interface.py:140-148)
  mamba_inner_fn / mamba_inner_fn_no_out_proj            -- selective_scan_interface.py:608-638 (signatures), computing
        what mamba_inner_ref (:640-674) computes: conv+SiLU -> x_proj -> dt_proj -> scan(z gate) [-> out_proj]

The projections are plain GEMMs and go to the library (torch.matmul = hipBLASLt); everything else is hand-written.
fp32 tensors, real A, d_state 16; anything else raises (no eager fallback).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from ._lib import call, ptr, stream_ptr
from .selective_scan import _prep, selective_scan_fn


class _CausalConv1dSiLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x, weight, bias = _prep(x, "x"), _prep(weight, "weight"), _prep(bias, "bias")
        Bt, D, L = x.shape
        W = weight.shape[-1]
        y = torch.empty_like(x)
        call("nnz_causal_conv1d_silu_forward", ptr(x), ptr(weight), ptr(bias), ptr(y), Bt, D, L, W, stream_ptr())
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        Bt, D, L = x.shape
        W = weight.shape[-1]
        dy = dy.float().contiguous()
        dx = torch.empty_like(x)
        dw = torch.empty_like(weight)
        db = torch.empty_like(bias) if bias is not None else None
        call("nnz_causal_conv1d_silu_backward", ptr(x), ptr(weight), ptr(bias), ptr(dy), ptr(dx), ptr(dw), ptr(db), Bt, D,
             L, W, stream_ptr())
        return dx, dw, db


def causal_conv1d_fn(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                     activation: Optional[str] = None) -> torch.Tensor:
    """x (B, D, L), weight (D, W) [or the nn.Conv1d layout (D, 1, W)], bias (D); activation must be silu / swish"""
    if activation not in ("silu", "swish"):
        raise NotImplementedError("causal_conv1d_fn: the Mamba block uses activation='silu' (swish); got %r" % activation)
    if weight.dim() == 3:
        weight = weight.squeeze(1)
    if weight.shape[-1] > 8:
        raise NotImplementedError("causal_conv1d_fn: kernel width <= 8")
    return _CausalConv1dSiLU.apply(x, weight, bias)


class _SiluGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, z):
        y, z = _prep(y, "y"), _prep(z, "z")
        out = torch.empty_like(y)
        call("nnz_silu_gate_forward", ptr(y), ptr(z), ptr(out), y.numel(), stream_ptr())
        ctx.save_for_backward(y, z)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, z = ctx.saved_tensors
        dout = dout.float().contiguous()
        dy, dz = torch.empty_like(y), torch.empty_like(z)
        call("nnz_silu_gate_backward", ptr(dout), ptr(y), ptr(z), ptr(dy), ptr(dz), y.numel(), stream_ptr())
        return dy, dz


def silu_gate(y: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    return _SiluGate.apply(y, z)


def mamba_inner_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B=None, C=None,
                               D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
    """xz (B, 2*d_inner, L) -> gated scan output (B, d_inner, L); input-dependent B / C only (the reference's call
    sites pass B=None, C=None)."""
    if B is not None or C is not None:
        raise NotImplementedError("mamba_inner_fn: constant B / C are not used by the reference's Mamba module")
    if A.is_complex():
        raise NotImplementedError("mamba_inner_fn: complex A is not supported")
    Bt, _, L = xz.shape
    R = delta_proj_weight.shape[1]
    N = A.shape[-1]
    x, z = xz.chunk(2, dim=1)
    x = causal_conv1d_fn(x, conv1d_weight, conv1d_bias, "silu")
    x_dbl = F.linear(x.transpose(1, 2).reshape(Bt * L, -1), x_proj_weight)          # (B L, R + 2N)
    delta = (delta_proj_weight @ x_dbl[:, :R].t()).view(-1, Bt, L).transpose(0, 1)   # (B, d_inner, L)
    Bm = x_dbl[:, R:R + N]
    Cm = x_dbl[:, R + N:R + 2 * N]
    if B_proj_bias is not None:
        Bm = Bm + B_proj_bias.to(Bm.dtype)
    if C_proj_bias is not None:
        Cm = Cm + C_proj_bias.to(Cm.dtype)
    Bm = Bm.view(Bt, L, N).transpose(1, 2).contiguous()
    Cm = Cm.view(Bt, L, N).transpose(1, 2).contiguous()
    return selective_scan_fn(x, delta, A, Bm, Cm, D, z=z, delta_bias=delta_bias, delta_softplus=delta_softplus)


def mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias, A,
                   B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
    y = mamba_inner_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D,
                                   delta_bias, B_proj_bias, C_proj_bias, delta_softplus)
    return F.linear(y.transpose(1, 2), out_proj_weight, out_proj_bias)
