"""Window-attention core as an autograd Function over the gfx950 MFMA kernels (csrc/window_attention.hip).

`window_attention_core(qkv, bias_table, bias_index_i32, num_heads, shift_size, scale)` computes, for a (B, H, W, 3C)
qkv tensor, what WindowAttention.forward of the reference does between its qkv and proj Linears
(/root/reference/nnunetv2/nets/swt2net.py:584-619): roll, 7x7 window partition, per-head
softmax(q*scale @ k^T + table[index] (+ -100 region mask)) @ v, head merge, un-partition, roll back -> (B, H, W, C).
The relative-position bias is looked up (and its gradient scattered) inside the kernels.
fp32 only (the reference's Swin trainers run without autocast, nnUNetTrainerSwT2Net.py:112-130); CPU tensors raise.
"""
from __future__ import annotations

import torch

from ._lib import call, ptr, stream_ptr


class WindowAttentionCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv: torch.Tensor, bias_table: torch.Tensor, bias_index: torch.Tensor, num_heads: int, shift: int,
                scale: float):
        if not qkv.is_cuda:
            raise RuntimeError("window_attention_core runs on MI355X through libnnuzoo_hip.so only (no CPU fallback)")
        if bias_index.dtype != torch.int32 or not bias_index.is_contiguous():
            raise ValueError("bias_index must be a contiguous int32 tensor")
        qkv = qkv.float().contiguous()
        table = bias_table.float().contiguous()
        B, H, W, C3 = qkv.shape
        C = C3 // 3
        out = torch.empty((B, H, W, C), dtype=torch.float32, device=qkv.device)
        call("nnz_window_attention_forward", ptr(qkv), ptr(table), ptr(bias_index), ptr(out), B, H, W, C, num_heads,
             shift, float(scale), stream_ptr())
        ctx.save_for_backward(qkv, table, bias_index)
        ctx.cfg = (B, H, W, C, num_heads, shift, float(scale))
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, table, bias_index = ctx.saved_tensors
        B, H, W, C, heads, shift, scale = ctx.cfg
        dout = dout.float().contiguous()
        dqkv = torch.empty_like(qkv)
        dtable = torch.empty_like(table)
        call("nnz_window_attention_backward", ptr(qkv), ptr(table), ptr(bias_index), ptr(dout), ptr(dqkv), ptr(dtable),
             B, H, W, C, heads, shift, scale, stream_ptr())
        return dqkv, dtable, None, None, None, None


def window_attention_core(qkv, bias_table, bias_index_i32, num_heads: int, shift: int, scale: float):
    return WindowAttentionCore.apply(qkv, bias_table, bias_index_i32, num_heads, shift, scale)
