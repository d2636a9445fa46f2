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

_CHECKED_INDEX = set()


def _check_index_layout(bias_index: torch.Tensor) -> None:
    """the backward kernel sums the bias-table gradient by displacement, index[i][j] = (yi - yj + 6) * 13 + (xi - xj + 6) -
    the layout the reference registers (swt2net.py:545).  Verified once per buffer; any other content is refused loudly."""
    key = (bias_index.data_ptr(), bias_index._version)
    if key in _CHECKED_INDEX:
        return
    ar = torch.arange(7)
    yy, xx = torch.meshgrid(ar, ar, indexing="ij")
    y, x = yy.flatten(), xx.flatten()
    expect = (y[:, None] - y[None, :] + 6) * 13 + (x[:, None] - x[None, :] + 6)
    if bias_index.shape != (49, 49) or not torch.equal(bias_index.cpu().long(), expect):
        raise NotImplementedError("window_attention_core: relative_position_index does not have the reference's "
                                  "displacement layout (swt2net.py:545)")
    _CHECKED_INDEX.add(key)


class WindowAttentionCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv: torch.Tensor, bias_table: torch.Tensor, bias_index: torch.Tensor, num_heads: int, shift: int,
                scale: float):
        if not qkv.is_cuda:
            raise RuntimeError("window_attention_core runs on MI355X through libnnuzoo_hip.so only (no CPU fallback)")
        if bias_index.dtype != torch.int32 or not bias_index.is_contiguous():
            raise ValueError("bias_index must be a contiguous int32 tensor")
        _check_index_layout(bias_index)
        qkv = qkv.float().contiguous()
        table = bias_table.float().contiguous()
        B, H, W, C3 = qkv.shape
        C = C3 // 3
        out = torch.empty((B, H, W, C), dtype=torch.float32, device=qkv.device)
        from .hip_ops import TIMER
        # algorithmic FLOPs: 4 L^2 hd per (window, head) = q k^T and p v, L = 49
        flops = 4.0 * 49 * 49 * (C // num_heads) * num_heads * B * (H // 7) * (W // 7)
        TIMER.wrap("win_attn_fwd", flops, lambda: call(
            "nnz_window_attention_forward", ptr(qkv), ptr(table), ptr(bias_index), ptr(out), B, H, W, C, num_heads,
            shift, float(scale), stream_ptr()))
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
        from .hip_ops import det_scratch
        # fixed-point sums of the bias-table gradient (deterministic) + one ticket record per head behind them
        sc = det_scratch(qkv.device, 170 * heads)
        from .hip_ops import TIMER
        # backward: dQ, dK, dV and dP = dO v^T are four more L^2 hd products (the recomputed q k^T is not counted)
        flops = 8.0 * 49 * 49 * (C // heads) * heads * B * (H // 7) * (W // 7)
        TIMER.wrap("win_attn_bwd", flops, lambda: call(
            "nnz_window_attention_backward", ptr(qkv), ptr(table), ptr(bias_index), ptr(dout), ptr(dqkv), ptr(dtable),
            ptr(sc.acc), ptr(sc.counter), B, H, W, C, heads, shift, scale, stream_ptr()))
        return dqkv, dtable, None, None, None, None


def window_attention_core(qkv, bias_table, bias_index_i32, num_heads: int, shift: int, scale: float):
    return WindowAttentionCore.apply(qkv, bias_table, bias_index_i32, num_heads, shift, scale)
