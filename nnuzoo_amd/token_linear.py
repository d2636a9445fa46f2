"""`TokenLinear`: nn.Linear (same parameters, state_dict, autocast numerics) for token-major activations with very many
tokens and few features - the in_proj / out_proj / patch merge / expand layers of the VSS blocks at 512^2 (524 288 tokens
x 16..128 features).  Forward and input gradient are the library GEMMs; the WEIGHT gradient dW = dY^T X is a (out x in)
matrix of a few thousand entries contracted over 10^5..10^6 tokens, for which the library picks a 16x16 macro-tile
without split-K and runs on a handful of workgroups (1 ms per call on MI355X; 10 such calls per M2Net step).  Here the
token axis is cut into chunks: one batched GEMM over strided views (no copies) plus a small sum."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

MIN_TOKENS = 65536
MAX_FEATURES = 256


def _chunks(T: int) -> int:
    nc = 1
    while nc < 128 and T % (nc * 2) == 0 and T // (nc * 2) >= 4096:
        nc *= 2
    return nc


class _TallLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        half = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.float16
        cd = torch.float16 if half else x.dtype
        xc = x if x.dtype == cd else x.to(cd)
        wc = weight if weight.dtype == cd else weight.to(cd)
        bc = None if bias is None else (bias if bias.dtype == cd else bias.to(cd))
        with torch.autocast("cuda", enabled=False):
            y = F.linear(xc, wc, bc)
        ctx.save_for_backward(xc, wc)
        ctx.meta = (x.dtype, weight.dtype, bias is not None, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        xdt, wdt, has_bias, bdt = ctx.meta
        fin, fout = wc.shape[1], wc.shape[0]
        with torch.autocast("cuda", enabled=False):
            dy2 = dy.to(wc.dtype).reshape(-1, fout)
            x2 = xc.reshape(-1, fin)
            dx = dw = db = None
            if ctx.needs_input_grad[0]:
                dx = (dy2 @ wc).view(xc.shape).to(xdt)
            if ctx.needs_input_grad[1]:
                T = x2.shape[0]
                nc = _chunks(T)
                if nc > 1:
                    part = torch.bmm(dy2.view(nc, T // nc, fout).transpose(1, 2), x2.view(nc, T // nc, fin))
                    dw = part.sum(0, dtype=torch.float32).to(wdt)
                else:
                    dw = (dy2.t() @ x2).to(wdt)
            if has_bias and ctx.needs_input_grad[2]:
                db = dy2.sum(0, dtype=torch.float32).to(bdt)
        return dx, dw, db


class TokenLinear(nn.Linear):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        tokens = x.numel() // max(1, x.shape[-1])
        if x.is_cuda and tokens >= MIN_TOKENS and self.in_features <= MAX_FEATURES and self.out_features <= MAX_FEATURES \
                and x.is_contiguous() and x.dtype in (torch.float16, torch.float32):
            return _TallLinearFn.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)
