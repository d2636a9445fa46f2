"""The Swin transformer block as ONE autograd node of five forward and seven backward launches (fp32, gfx950).

Reference: `SwinTransformerBlock.forward`, /root/reference/nnunetv2/nets/swt2net.py:622-661 (pad top / left to a multiple of the 7-token
window, norm1, `WindowAttention.forward` :584-619, DropPath :379-388 + residual, norm2, `Mlp` :496-515, DropPath + residual, crop).
Module by module that is 11 launches forward and 11 backward, six of them (pad, two LayerNorms, two residual adds, crop) with
nothing to do but move the activation once more - and 144 such blocks run per SwT2Net step, most of them on a few hundred to a
few thousand tokens where a launch costs more than its work (profiles/r05_swin_ops_before.txt).  Here:

  forward   K1  [pad gather + LayerNorm + qkv Linear]      csrc/dense32.hip, LN prologue; padded tokens are zero rows -> beta
            K2  window attention                           csrc/window_attention.hip; qkv on the padded grid, output on the unpadded
            K3  [proj Linear + DropPath + residual]        epilogue: x1 = x + s_b (a Wp^T + bp)   (the crop happened in K2's addressing)
            K4  [LayerNorm + fc1 + GELU]                   LN prologue, dual output (pre-activation kept for GELU')
            K5  [fc2 + DropPath + residual]
  backward  B1  dh = s_b (dx2 W2) * GELU'(h)      B2  dn2 = dh W1      B3  dx1 = LayerNorm-backward(dn2) + dx2
            B4  dao = s_b (dx1 Wp)                B5  window attention backward (dout read on the unpadded grid, zeros elsewhere)
            B6  dn1 = dqkv Wqkv                   B7  dx = crop(LayerNorm-backward(dn1)) + dx1
            the four weight gradients (DropPath folded into their dy operand) and the two LayerNorm dgamma | dbeta folds are queued for
            the pass's ONE grouped launch (token_linear.deferred_wgrads).

The MLP half and the projection run on the UNPADDED tokens (the reference computes them on the padded grid and crops: 2.3 x the
tokens at the 16^2 level, 3 x at 8^2).  Same arithmetic per token as the module-by-module path; LayerNorm statistics are formed
by the consuming GEMM's workgroups (two passes over the row, like csrc/layer_norm.hip).
"""
from __future__ import annotations

import os

import torch

from . import _lib
from . import backends as _backends
from ._lib import call, ptr, stream_ptr
from .hip_ops import det_scratch

USE_FUSED_BLOCK = os.environ.get("NNZ_SWIN_FUSED", "1") != "0"
_WS = {}
_WS_OUTGROWN = []      # a captured hipGraph keeps the ADDRESS of the buffer it was recorded with: outgrown buffers stay alive


def _workspace(device, floats: int):
    """split-K partials of the skinny products: one buffer per (device, STREAM).  The partials are written by one launch and read by
    the fold launch behind it, so calls on ONE stream may share a buffer; two streams (an eager validation pass beside a captured
    step's side stream, GraphedDDPStep segments) must not.  A capture runs on a stream of its own, so its buffer is
    allocated inside that capture from the graph's pool and is never handed to eager code; buffers that were outgrown stay alive
    because a captured graph keeps the ADDRESS it was recorded with."""
    if floats <= 0:
        return None
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < floats:
        if torch.cuda.is_current_stream_capturing() and ws is not None:
            raise RuntimeError("swin_block: the split-K workspace cannot grow during a capture - run an eager pass first")
        if ws is not None:
            _WS_OUTGROWN.append(ws)
        ws = torch.empty(max(floats, 1 << 22), dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


def _fwd(x, w, b, y, y_act, T, K, N, gelu=0, ln=None, pad=None, res=None, dp=None):
    """one nnz_dense32_forward_fused call; ln = (gamma, beta, eps, mean, rstd, y) or None; pad = (H, W, py, px) or None;
    dp = (rand, keep, rows_per_sample, samples) or None"""
    ws = _workspace(x.device, int(_lib.load().nnz_dense32_splitk_workspace_floats(T, K, N)))
    g, bt, eps, mean, rstd, ly = ln if ln is not None else (None, None, 0.0, None, None, None)
    ph, pw, py, px = pad if pad is not None else (0, 0, 0, 0)
    r, keep, rps, nb = dp if dp is not None else (None, 1.0, 1, 1)
    call("nnz_dense32_forward_fused", ptr(x), ptr(w), ptr(b), ptr(y), ptr(y_act), T, K, N, int(gelu), ptr(g), ptr(bt), float(eps),
         ptr(mean), ptr(rstd), ptr(ly), ph, pw, py, px, ptr(res), ptr(r), float(keep), int(rps), int(nb), ptr(ws), stream_ptr())


def _dgrad(dy, w, h, dx, T, K, N, dp=None):
    ws = _workspace(dy.device, int(_lib.load().nnz_dense32_splitk_workspace_floats(T, N, K)))
    r, keep, rps, nb = dp if dp is not None else (None, 1.0, 1, 1)
    call("nnz_dense32_dgrad_fused", ptr(dy), ptr(w), ptr(h), ptr(dx), T, K, N, ptr(r), float(keep), int(rps), int(nb), ptr(ws),
         stream_ptr())


def _deferrable(params) -> bool:
    from .token_linear import _DEFER, _has_grad_hooks
    return _DEFER["on"] and all(p is None or (p.is_leaf and p.requires_grad and not _has_grad_hooks(p)) for p in params)


class _SwinBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n1w, n1b, qkvw, qkvb, table, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b, draws1, draws2, cfg):
        heads, shift, scale, keep, eps1, eps2, idx32 = cfg
        B, H, W, C = x.shape
        x = x.contiguous()
        dev = x.device
        padded = H % 7 != 0 or W % 7 != 0
        py, px = (7 - H % 7, 7 - W % 7) if padded else (0, 0)      # a full extra window on an axis that divides (:643-645)
        Hp, Wp = H + py, W + px
        T, Tp = B * H * W, B * Hp * Wp
        f32 = dict(dtype=torch.float32, device=dev)
        dp1 = (draws1, keep, H * W, B) if draws1 is not None else None
        dp2 = (draws2, keep, H * W, B) if draws2 is not None else None
        qkv, n1 = torch.empty((Tp, 3 * C), **f32), torch.empty((Tp, C), **f32)
        mean1, rstd1 = torch.empty(Tp, **f32), torch.empty(Tp, **f32)
        _fwd(x, qkvw, qkvb, qkv, None, Tp, C, 3 * C, ln=(n1w, n1b, eps1, mean1, rstd1, n1), pad=(H, W, py, px) if padded else None)
        ao = torch.empty((T, C), **f32)
        from .hip_ops import TIMER
        # algorithmic FLOPs of the attention core: 4 L^2 hd per (window, head), L = 49 - bench.py's roofline record
        TIMER.wrap("win_attn_fwd", 4.0 * 49 * 49 * C * B * (Hp // 7) * (Wp // 7), lambda: call(
            "nnz_window_attention_forward_pad", ptr(qkv), ptr(table), ptr(idx32), ptr(ao), B, Hp, Wp, C, heads, shift, float(scale),
            py, px, stream_ptr()))
        x1 = torch.empty((T, C), **f32)
        _fwd(ao, projw, projb, x1, None, T, C, C, res=x, dp=dp1)
        Hd = fc1w.shape[0]
        h, act, n2 = torch.empty((T, Hd), **f32), torch.empty((T, Hd), **f32), torch.empty((T, C), **f32)
        mean2, rstd2 = torch.empty(T, **f32), torch.empty(T, **f32)
        _fwd(x1, fc1w, fc1b, h, act, T, C, Hd, gelu=1, ln=(n2w, n2b, eps2, mean2, rstd2, n2))
        x2 = torch.empty((T, C), **f32)
        _fwd(act, fc2w, fc2b, x2, None, T, Hd, C, res=x1, dp=dp2)
        ctx.save_for_backward(x, qkv, n1, mean1, rstd1, ao, x1, n2, mean2, rstd2, h, act, draws1, draws2, table, idx32)
        ctx.params = (n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b)
        ctx.table_param = table
        ctx.geo = (B, H, W, C, py, px, heads, shift, scale, keep, padded)
        return x2.view(B, H, W, C)

    @staticmethod
    def backward(ctx, dx2):
        from .token_linear import _DEFER, _d32_backward_products, defer_fold
        x, qkv, n1, mean1, rstd1, ao, x1, n2, mean2, rstd2, h, act, draws1, draws2, table, idx32 = ctx.saved_tensors
        n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b = ctx.params
        B, H, W, C, py, px, heads, shift, scale, keep, padded = ctx.geo
        Hp, Wp = H + py, W + px
        T, Tp = B * H * W, B * Hp * Wp
        Hd = fc1w.shape[0]
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        dx2 = dx2.reshape(T, C)
        if not dx2.is_contiguous() or dx2.dtype != torch.float32:
            dx2 = dx2.float().contiguous()
        dp1 = (draws1, keep, H * W, B) if draws1 is not None else None
        dp2 = (draws2, keep, H * W, B) if draws2 is not None else None
        ni = ctx.needs_input_grad
        deferred = _deferrable(ctx.params)
        grads = {}

        def wgrad(dy2, x2, w, b, dp, iw, ib):
            """weight / bias gradient of one Linear: queued for the grouped launch, or computed here (parity tests, hooks)"""
            if not (ni[iw] or (b is not None and ni[ib])):
                return
            if deferred:
                _DEFER["jobs"].append((dy2, x2, w, b if (b is not None and ni[ib]) else None, dp))
                return
            if dp is not None:                     # the scaled dy as a tensor of its own (this path is not the hot one)
                s = torch.floor(dp[0].reshape(-1) + dp[1]) / dp[1]
                dy2 = (dy2.view(dp[3], -1) * s.view(-1, 1)).view_as(dy2)
            _, dw, db = _d32_backward_products(dy2, x2, w, False, ni[iw], b is not None and ni[ib], bias=b)
            grads[iw], grads[ib] = dw, db

        def ln_fold(part, parts, gw, gb, iw, ib):
            def assign(dst):
                for p, g in ((gw, dst[:C]), (gb, dst[C:])):
                    if p is None:
                        continue
                    if p.grad is None:
                        p.grad = g
                    else:
                        p.grad.add_(g)
            if deferred and defer_fold(part, 2 * C, parts, assign):
                return
            s = part.view(parts, 2 * C).sum(0)
            grads[iw], grads[ib] = (s[:C] if gw is not None else None), (s[C:] if gb is not None else None)

        # ---- MLP half ---------------------------------------------------------------------------------------------------------------
        dh = torch.empty((T, Hd), **f32)
        _dgrad(dx2, fc2w, h, dh, T, Hd, C, dp=dp2)
        wgrad(dx2, act, fc2w, fc2b, dp2, 12, 13)
        dn2 = torch.empty((T, C), **f32)
        _dgrad(dh, fc1w, None, dn2, T, C, Hd)
        wgrad(dh, n2, fc1w, fc1b, None, 10, 11)
        lib = _lib.load()
        parts2 = int(lib.nnz_layer_norm_backward_parts(T, C))
        part2 = torch.empty((parts2, 2 * C), **f32)
        dx1 = torch.empty((T, C), **f32)
        call("nnz_layer_norm_backward_partial", ptr(x1), ptr(n2w), ptr(mean2), ptr(rstd2), ptr(dn2), ptr(dx2), ptr(dx1), ptr(part2),
             T, C, 0, 0, 0, 0, stream_ptr())
        ln_fold(part2, parts2, n2w, n2b, 8, 9)
        # ---- attention half ---------------------------------------------------------------------------------------------------------
        dao = torch.empty((T, C), **f32)
        _dgrad(dx1, projw, None, dao, T, C, C, dp=dp1)
        wgrad(dx1, ao, projw, projb, dp1, 6, 7)
        dqkv = torch.empty((Tp, 3 * C), **f32)
        dtable = None
        tparam = ctx.table_param
        from .hip_ops import TIMER
        aflops = 8.0 * 49 * 49 * C * B * (Hp // 7) * (Wp // 7)    # dQ, dK, dV, dP (the recomputed q k^T is not counted)
        if ni[5] and deferred and _deferrable((tparam,)):
            # the bias-table gradient leaves the launch as per-workgroup shares; the pass's grouped launch folds them
            tparts = int(lib.nnz_window_attention_backward_parts(B, Hp, Wp, heads))
            tpart = torch.empty((tparts, 169 * heads), **f32)
            TIMER.wrap("win_attn_bwd", aflops, lambda: call(
                "nnz_window_attention_backward_partial", ptr(qkv), ptr(table), ptr(idx32), ptr(dao), ptr(dqkv), ptr(tpart), B, Hp,
                Wp, C, heads, shift, float(scale), py, px, stream_ptr()))

            def assign_table(dst):
                g = dst.view(169, heads)
                if tparam.grad is None:
                    tparam.grad = g
                else:
                    tparam.grad.add_(g)
            defer_fold(tpart, 169 * heads, tparts, assign_table)
        else:
            dtable = torch.empty_like(table)
            sc = det_scratch(dev, 170 * heads)
            TIMER.wrap("win_attn_bwd", aflops, lambda: call(
                "nnz_window_attention_backward_pad", ptr(qkv), ptr(table), ptr(idx32), ptr(dao), ptr(dqkv), ptr(dtable),
                ptr(sc.acc), ptr(sc.counter), B, Hp, Wp, C, heads, shift, float(scale), py, px, stream_ptr()))
        dn1 = torch.empty((Tp, C), **f32)
        _dgrad(dqkv, qkvw, None, dn1, Tp, C, 3 * C)
        wgrad(dqkv, n1, qkvw, qkvb, None, 3, 4)
        parts1 = int(lib.nnz_layer_norm_backward_parts(Tp, C))
        part1 = torch.empty((parts1, 2 * C), **f32)
        dx = torch.empty((T, C), **f32)
        call("nnz_layer_norm_backward_partial", ptr(x), ptr(n1w), ptr(mean1), ptr(rstd1), ptr(dn1), ptr(dx1), ptr(dx), ptr(part1),
             Tp, C, H if padded else 0, W if padded else 0, py, px, stream_ptr())
        ln_fold(part1, parts1, n1w, n1b, 1, 2)
        out = [None] * 17
        out[0] = dx.view(B, H, W, C) if ni[0] else None
        out[5] = dtable if (ni[5] and dtable is not None) else None
        for k, g in grads.items():
            out[k] = g if (g is not None and ni[k]) else None
        return tuple(out)


def fused_block_ok(blk, x: torch.Tensor) -> bool:
    """fp32 device step without autocast (the reference's Swin trainers, nnUNetTrainerSwT2Net.py:112-130), the block's dropouts at 0,
    exact GELU, head dim within the attention kernel's range"""
    from torch import nn
    if not (USE_FUSED_BLOCK and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4) or torch.is_autocast_enabled():
        return False
    a, m = blk.attn, blk.mlp
    C = x.shape[-1]
    hd = C // a.num_heads
    if C % 4 or C != a.num_heads * hd or hd % 2 or hd > 32 or hd < 2:
        return False
    if a.proj_drop.p or a.attn_drop.p or m.drop1.p or m.drop2.p:
        return False
    if not (isinstance(m.act, nn.GELU) and m.act.approximate == "none"):
        return False
    for lin in (a.qkv, a.proj, m.fc1, m.fc2):
        if lin.weight.dtype != torch.float32 or lin.bias is None:
            return False
    if len(blk.norm1.normalized_shape) != 1 or blk.norm1.weight is None or blk.norm2.weight is None:
        return False
    dp = blk.drop_path
    if hasattr(dp, "drop_prob") and blk.training and dp.drop_prob >= 1.0:
        return False
    return x.shape[0] * (x.shape[1] + 7) <= 1 << 24


def swin_block_forward(blk, x: torch.Tensor) -> torch.Tensor:
    """`SwinTransformerBlock.forward` on the fused node; the per-sample DropPath draws are made exactly like the reference's
    DropPath (B uniform draws per branch, attention branch first: droppath_draws.uniform, the same call the module path makes)"""
    a, m, dp = blk.attn, blk.mlp, blk.drop_path
    B = x.shape[0]
    draws1 = draws2 = None
    keep = 1.0
    if hasattr(dp, "drop_prob") and dp.drop_prob > 0. and blk.training:
        keep = 1.0 - dp.drop_prob
        from .droppath_draws import uniform      # rows of the pass's draw table (one launch per forward) when one is open
        draws1 = uniform(B, x.device)
        draws2 = uniform(B, x.device)
    for lin in (a.qkv, a.proj, m.fc1, m.fc2):
        _backends.note(lin, "hip-f32")
    cfg = (a.num_heads, a.shift_size, a.scale, keep, blk.norm1.eps, blk.norm2.eps, a._index_i32())
    return _SwinBlockFn.apply(x, blk.norm1.weight, blk.norm1.bias, a.qkv.weight, a.qkv.bias, a.relative_position_bias_table,
                              a.proj.weight, a.proj.bias, blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias,
                              m.fc2.weight, m.fc2.bias, draws1, draws2, cfg)
