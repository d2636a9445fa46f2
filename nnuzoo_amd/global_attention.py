"""Global multi-head self-attention core of the ViT blocks (monai SABlock as used by UNETR, reference binding
/root/reference/nnunetv2/nets/unetr2net.py:10,1414-1428): softmax(q k^T * scale) v over all L <= 1024 patch tokens, head_dim
8 / 16 / 32, from the packed qkv projection (B, L, 3, heads, head_dim) to the merged-head output (B, L, heads * head_dim).

The score matrix is never written to HBM: the contraction runs through the fused scaled-dot-product kernels of the ROCm
stack (`torch.nn.functional.scaled_dot_product_attention`, flash / memory-efficient back ends), reading q, k, v as strided
views of the qkv tensor.  (A hand-written kernel for this shape class is the same problem as csrc/window_attention.hip
with 1024 keys instead of 49 - listed as next in DESIGN.md.)"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def global_attention(qkv: torch.Tensor, scale: float) -> torch.Tensor:
    B, L, three, H, D = qkv.shape
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))          # (B, H, L, D) views
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=scale)
    return o.transpose(1, 2).reshape(B, L, H * D)
