"""Global multi-head self-attention core of the ViT blocks (monai SABlock as used by UNETR, reference binding
/root/reference/nnunetv2/nets/unetr2net.py:10,1414-1428): softmax(q k^T * scale) v over all L <= 1024 patch tokens, head_dim
8 / 16 / 32, from the packed qkv projection (B, L, 3, heads, head_dim) to the merged-head output (B, L, heads * head_dim).

Hand-written flash-style kernels (csrc/global_attention.hip, fp32 MFMA; round 2 forwarded to torch's SDPA): the score
matrix is never written to HBM, the backward is two atomic-free passes (keys side: dK / dV, queries side: dQ) that
recompute P from the saved log-sum-exp.  fp16 inputs (autocast steps) are widened to fp32 operands: products of fp16
values are exact in fp32 and the accumulation is fp32 either way, so the result is at least as accurate as the
reference's autocast einsum / softmax / einsum chain; the output takes the input's dtype.  CPU tensors raise."""
from __future__ import annotations

import torch

from ._lib import call, ptr, stream_ptr


class _GlobalAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv: torch.Tensor, scale: float):
        if not qkv.is_cuda:
            raise RuntimeError("global_attention runs on MI355X through libnnuzoo_hip.so only (no CPU fallback)")
        B, L, three, H, D = qkv.shape
        if three != 3 or D % 2 or D > 32:
            raise NotImplementedError(f"global_attention: head_dim must be even and <= 32, got {D}")
        q32 = qkv.float().contiguous()
        out = torch.empty((B, L, H * D), dtype=torch.float32, device=qkv.device)
        lse = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
        call("nnz_global_attention_forward", ptr(q32), ptr(out), ptr(lse), B, L, H, D, float(scale), stream_ptr())
        ctx.save_for_backward(q32, out, lse)
        ctx.cfg = (B, L, H, D, float(scale), qkv.dtype)
        return out if qkv.dtype == torch.float32 else out.to(qkv.dtype)

    @staticmethod
    def backward(ctx, dout):
        q32, out, lse = ctx.saved_tensors
        B, L, H, D, scale, dt = ctx.cfg
        dout = dout.float().contiguous()
        dqkv = torch.empty_like(q32)
        call("nnz_global_attention_backward", ptr(q32), ptr(out), ptr(lse), ptr(dout), ptr(dqkv), B, L, H, D, scale,
             stream_ptr())
        return (dqkv if dt == torch.float32 else dqkv.to(dt)), None


def hip_supported(qkv: torch.Tensor) -> bool:
    """what csrc/global_attention.hip takes: a device tensor with an even head_dim <= 32 (the zoo's ViT stages: 8 / 16 / 32)"""
    return qkv.is_cuda and qkv.dim() == 5 and qkv.shape[2] == 3 and qkv.shape[4] % 2 == 0 and qkv.shape[4] <= 32


def global_attention(qkv: torch.Tensor, scale: float, module: torch.nn.Module = None) -> torch.Tensor:
    """qkv: (B, L, 3, heads, head_dim) -> (B, L, heads * head_dim).  Shapes outside the kernel's set (monai's UNETR / ViT
    defaults hidden 768 / 12 heads = head_dim 64) and CPU tensors (structure checks, the planner) run on torch's
    scaled_dot_product_attention; the choice is recorded on `module` (nnuzoo_amd/backends.py) - never silent."""
    if hip_supported(qkv):
        if module is not None:
            from . import backends
            backends.note(module, "hip-f32", "attention")
        return _GlobalAttentionFn.apply(qkv, scale)
    if module is not None:
        from . import backends
        backends.note(module, "library", "attention", why=f"head_dim {qkv.shape[-1]} on {qkv.device.type}")
    B, L, _, H, D = qkv.shape
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))           # (B, H, L, D)
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=scale)
    return o.transpose(1, 2).reshape(B, L, H * D)
