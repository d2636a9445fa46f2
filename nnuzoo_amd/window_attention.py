"""Window-attention core as an autograd Function over the gfx950 MFMA kernels (csrc/window_attention.hip).

`window_attention_core(qkv, bias_dense, num_heads, shift_size, scale)` computes, for a (B, H, W, 3C) qkv tensor,
what WindowAttention.forward of the reference does between its qkv and proj Linears
(/root/reference/nnunetv2/nets/swt2net.py:584-619): roll, 7x7 window partition, per-head
softmax(q*scale @ k^T + bias (+ -100 region mask)) @ v, head merge, un-partition, roll back -> (B, H, W, C).
fp32 only (the reference's Swin trainers run without autocast, nnUNetTrainerSwT2Net.py:112-130); CPU tensors raise.
"""
from __future__ import annotations

import torch

from ._lib import call, ptr, stream_ptr


class WindowAttentionCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv: torch.Tensor, bias_dense: torch.Tensor, num_heads: int, shift: int, scale: float):
        if not qkv.is_cuda:
            raise RuntimeError("window_attention_core runs on MI355X through libnnuzoo_hip.so only (no CPU fallback)")
        qkv = qkv.float().contiguous()
        bias_dense = bias_dense.float().contiguous()
        B, H, W, C3 = qkv.shape
        C = C3 // 3
        out = torch.empty((B, H, W, C), dtype=torch.float32, device=qkv.device)
        call("nnz_window_attention_forward", ptr(qkv), ptr(bias_dense), ptr(out), B, H, W, C, num_heads, shift,
             float(scale), stream_ptr())
        ctx.save_for_backward(qkv, bias_dense)
        ctx.cfg = (B, H, W, C, num_heads, shift, float(scale))
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, bias_dense = ctx.saved_tensors
        B, H, W, C, heads, shift, scale = ctx.cfg
        dout = dout.float().contiguous()
        dqkv = torch.empty_like(qkv)
        dbias = torch.empty_like(bias_dense)
        call("nnz_window_attention_backward", ptr(qkv), ptr(bias_dense), ptr(dout), ptr(dqkv), ptr(dbias), B, H, W, C,
             heads, shift, scale, stream_ptr())
        return dqkv, dbias, None, None, None


def window_attention_core(qkv, bias_dense, num_heads: int, shift: int, scale: float):
    return WindowAttentionCore.apply(qkv, bias_dense, num_heads, shift, scale)
