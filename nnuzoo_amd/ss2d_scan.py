"""`ss2d_cross_scan`: the core of the SS2D block (reference: SS2D.forward_core + the direction sum of SS2D.forward,
/root/reference/nnunetv2/nets/m2net.py:170-206, 217-219) as ONE autograd node on hand-written gfx950 kernels.

The reference materialises the four scan directions (stack of x and x^T, flip, cat), projects each copy with einsum,
expands dt with a second einsum, scans, and flips / transposes / adds the results back - about 30 element-wise and copy
launches forward and twice that backward per block, each a full pass over (B, 4 Di, L).  Here the directions are index
arithmetic inside the scan kernels (csrc/selective_scan.hip, cross-scan mode): one transposing copy of the input, two
small batched GEMMs (x_proj forward, its two gradients backward: library GEMMs), the chunk scan that forms delta from the
R dt rows on the fly, and one merge kernel that sums the directions into the token-major (B, H, W, Di) result.
fp32 throughout, as the reference forces for the scan (`.float()` at m2net.py:185-191).
"""
from __future__ import annotations

import os

import torch

from ._lib import call, load, ptr, stream_ptr
from .hip_ops import TIMER

N_STATE = 16
MAX_DT_RANK = 8


_USE_XPROJ = os.environ.get("NNZ_XPROJ", "1") != "0"   # A/B switch for measurements


def _xproj_ok(Di: int, C2: int) -> bool:
    """shapes served by csrc/ss2d_xproj.hip (32-channel weight chunks, at most 80 projection rows)"""
    return _USE_XPROJ and Di % 32 == 0 and 32 <= Di <= 1024 and 8 <= C2 <= 80


def _proj_weight_grad(dP: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """sum_{b,l} dP[s,b,c,l] x2[s,b,d,l].  For long sequences the library GEMM sees a 68 x 32 output with K = 10^5..10^6
    and runs on ~10 workgroups (1 ms per call at 512^2); cutting L into chunks makes it a batched GEMM over strided
    views (no copies) plus a small sum."""
    S, B, C2, L = dP.shape
    Di = x2.shape[2]
    nc = 1
    while nc < 64 and L % (nc * 2) == 0 and L // (nc * 2) >= 4096:
        nc *= 2
    if nc == 1:
        return torch.einsum("sbcl,sbdl->scd", dP, x2)
    lc = L // nc
    out = torch.zeros((S, C2, Di), dtype=dP.dtype, device=dP.device)
    for s_ in range(S):
        acc = None
        for b in range(B):
            a = dP[s_, b].view(C2, nc, lc).transpose(0, 1)                       # (nc, C2, lc), rows L apart
            x = x2[s_, b].view(Di, nc, lc).permute(1, 2, 0)                      # (nc, lc, Di)
            part = torch.bmm(a, x)                                              # (nc, C2, Di)
            acc = part if acc is None else acc + part
        out[s_] = acc.sum(0)
    return out


class _PrepareFn(torch.autograd.Function):
    """(B, Di, H, W) f16|f32 -> x2 [2][B][Di][L] f32: row-major and column-major token order (one transposing copy)"""

    @staticmethod
    def forward(ctx, xc):
        B, Di, H, W = xc.shape
        xc = xc.contiguous()
        x2 = torch.empty((2, B, Di, H * W), dtype=torch.float32, device=xc.device)
        call("nnz_ss2d_prepare", ptr(xc), int(xc.dtype == torch.float16), ptr(x2), B, Di, H, W, stream_ptr())
        ctx.meta = (B, Di, H, W, xc.dtype)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        B, Di, H, W, dt = ctx.meta
        dx2 = dx2.float().contiguous()
        dx = torch.empty((B, Di, H, W), dtype=dt, device=dx2.device)
        call("nnz_ss2d_merge_dx", 0, ptr(dx2), ptr(dx), int(dt == torch.float16), B, Di, H, W, stream_ptr())
        return dx


class _DwConvSiluPrepareFn(torch.autograd.Function):
    """depthwise 3x3 conv + SiLU on the token-major x half of the in_proj output, written straight as x2 (m2net.py:214
    `self.act(self.conv2d(x.permute(0, 3, 1, 2).contiguous()))` + the layout change of forward_core); csrc/ss2d_dwconv.hip"""

    @staticmethod
    def forward(ctx, x_tok, weight, bias):
        from .layer_norm import _row_stride
        B, H, W, Di = x_tok.shape
        xs = _row_stride(x_tok)
        if xs is None or x_tok.data_ptr() % 4:
            x_tok = x_tok.contiguous()
            xs = Di
        w9 = weight.detach().float().reshape(Di, 9)
        bv = None if bias is None else bias.detach().float()
        x2 = torch.empty((2, B, Di, H * W), dtype=torch.float32, device=x_tok.device)
        call("nnz_ss2d_dwconv_silu_forward", ptr(x_tok), int(x_tok.dtype == torch.float16), xs, ptr(w9), ptr(bv), ptr(x2),
             B, Di, H, W, stream_ptr())
        ctx.save_for_backward(x_tok, w9, bv)
        ctx.meta = (B, Di, H, W, xs, weight.shape)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        x_tok, w9, bv = ctx.saved_tensors
        B, Di, H, W, xs, wshape = ctx.meta
        dx2 = dx2.float().contiguous()
        dx = torch.empty((B, H, W, Di), dtype=x_tok.dtype, device=dx2.device)
        dwb = torch.empty(Di * 10, dtype=torch.float32, device=dx2.device)      # [Di][9] weights, then [Di] bias
        dw, db = dwb[:Di * 9], dwb[Di * 9:]
        from .token_linear import TWO_STAGE
        if TWO_STAGE:     # per-tile partial rows + fixed-order fold instead of fp32 atomics (bit-reproducible)
            nws = int(load().nnz_ss2d_dwconv_silu_backward_workspace_floats(B, Di, H, W))
            ws = torch.empty(nws, dtype=torch.float32, device=dx2.device)
            call("nnz_ss2d_dwconv_silu_backward_ws", ptr(x_tok), int(x_tok.dtype == torch.float16), xs, ptr(w9), ptr(bv),
                 ptr(dx2), ptr(dx), ptr(dw), ptr(db), ptr(ws), nws, B, Di, H, W, stream_ptr())
        else:
            call("nnz_ss2d_dwconv_silu_backward", ptr(x_tok), int(x_tok.dtype == torch.float16), xs, ptr(w9), ptr(bv),
                 ptr(dx2), ptr(dx), ptr(dw), ptr(db), B, Di, H, W, stream_ptr())
        return dx, dw.view(wshape), (db if bv is not None else None)


class _SS2DCrossScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2, hw, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
        lib = load()
        H, W = hw
        _, B, Di, L = x2.shape
        K = 4
        R = dt_projs_weight.shape[2]
        Cp = R + 2 * N_STATE
        f32 = dict(dtype=torch.float32, device=x2.device)
        with torch.autocast("cuda", enabled=False):
            # direction k = s + 2j  ->  rows [j*Cp, (j+1)*Cp) of source s's stacked projection weight
            if _xproj_ok(Di, 2 * Cp):
                # csrc/ss2d_xproj.hip: lanes = tokens; the kernels address the module's [4][Cp][Di] weight themselves (cp
                # argument) - no stacked copy per call, no un-stacking copy of its gradient
                Wst = x_proj_weight.detach().float().contiguous()
                P = torch.empty((2, B, 2 * Cp, L), **f32)
                call("nnz_ss2d_xproj_forward", ptr(x2), ptr(Wst), ptr(P), B, Di, 2 * Cp, L, Cp, stream_ptr())
            else:
                Wst = x_proj_weight.detach().float().view(2, 2, Cp, Di).transpose(0, 1).reshape(2, 1, 2 * Cp, Di)
                P = torch.matmul(Wst, x2)                                        # (2, B, 2Cp, L)
            A = A_logs.detach().float().contiguous()                             # A_log; the kernels use -exp(A_log)
            Wdt = dt_projs_weight.detach().float().reshape(K * Di, R).contiguous()
            bias = dt_projs_bias.detach().float().reshape(-1).contiguous()
            Dv = Ds.detach().float().contiguous()
            y = torch.empty((B, K, Di, L), **f32)
            state = torch.empty(lib.nnz_ss2d_scan_state_floats(B, Di, L), **f32)
            ws = torch.empty(lib.nnz_ss2d_scan_workspace_floats(B, Di, L), **f32)
            # algorithmic HBM bytes of the cross-scan forward: both sources of u once, the dt / B / C
            # rows of the four directions, y of the four directions
            TIMER.wrap("ss2d_scan_fwd", 4.0 * B * L * (2 * Di + 4 * Cp + 4 * Di), lambda: call(
                "nnz_ss2d_scan_forward", ptr(x2), ptr(P), ptr(Wdt), ptr(A), ptr(Dv), ptr(bias), ptr(y), ptr(state),
                ptr(ws), B, Di, R, L, 1, 1, stream_ptr()))
            out = torch.empty((B, H, W, Di), **f32)
            call("nnz_ss2d_merge", ptr(y), ptr(out), B, Di, H, W, stream_ptr())
        ctx.save_for_backward(x2, P, Wst, A, Wdt, bias, Dv, state)
        ctx.meta = (B, Di, H, W, R)
        ctx.xproj_param = x_proj_weight          # the parameter object itself: the deferred weight gradient sets its .grad
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = load()
        x2, P, Wst, A, Wdt, bias, Dv, state = ctx.saved_tensors
        B, Di, H, W, R = ctx.meta
        L, K = H * W, 4
        Cp = R + 2 * N_STATE
        f32 = dict(dtype=torch.float32, device=dout.device)
        with torch.autocast("cuda", enabled=False):
            dout = dout.float().contiguous()
            dy2 = torch.empty((2, B, Di, L), **f32)
            call("nnz_ss2d_split", ptr(dout), ptr(dy2), B, Di, H, W, stream_ptr())
            du = torch.empty((B, K, Di, L), **f32)
            dP = torch.empty_like(P)
            dWdt, dA = torch.empty_like(Wdt), torch.empty_like(A)
            dD, dbias = torch.empty_like(Dv), torch.empty_like(bias)
            gstate = torch.empty(lib.nnz_ss2d_scan_grad_state_floats(B, Di, L), **f32)
            ws = torch.empty(lib.nnz_ss2d_scan_workspace_floats(B, Di, L), **f32)
            # algorithmic bytes of the backward: reads u (2 sources), projections (4 dirs), dy (2 token orders); writes
            # du (4 dirs) and the projection gradient (4 dirs)
            TIMER.wrap("ss2d_scan_bwd", 4.0 * B * L * (2 * Di + 4 * Cp + 2 * Di + 4 * Di + 4 * Cp), lambda: call(
                "nnz_ss2d_scan_backward", ptr(x2), ptr(P), ptr(Wdt), ptr(A), ptr(Dv), ptr(bias), ptr(dy2), ptr(state),
                ptr(gstate), ptr(ws), ptr(du), ptr(dP), ptr(dWdt), ptr(dA), ptr(dD), ptr(dbias), B, Di, R, L, 1, 1,
                stream_ptr()))
            if _xproj_ok(Di, 2 * Cp):
                # W^T dP + the scans' own input gradients of the source's two directions, one pass (ss2d_xproj.hip)
                dx2 = torch.empty((2, B, Di, L), **f32)
                call("nnz_ss2d_xproj_backward_x", ptr(dP), ptr(Wst), ptr(du), ptr(dx2), B, Di, 2 * Cp, L, Cp, stream_ptr())
            else:
                dx2 = torch.matmul(Wst.transpose(-1, -2), dP)                    # (2, B, Di, L)
                # + the scans' own input gradients: direction k = 2j + s belongs to source s
                dx2 += du.view(B, 2, 2, Di, L).sum(1).transpose(0, 1)
            if _xproj_ok(Di, 2 * Cp) and L % 64 == 0 and ((2 * Cp + 7) // 8) * (Di // 8) <= 256:
                from .token_linear import TWO_STAGE, defer_xproj_wgrad
                if ctx.needs_input_grad[2] and defer_xproj_wgrad(dP, x2, ctx.xproj_param):
                    d_xproj = None      # queued: the pass's grouped launch writes x_proj_weight.grad (token_linear._XpKind)
                elif TWO_STAGE:
                    d_xproj = torch.empty((K, Cp, Di), **f32)                    # written in the module's layout
                    nws = int(lib.nnz_ss2d_xproj_backward_w_workspace_floats(B, Di, 2 * Cp, L))
                    ws = torch.empty(nws, **f32)
                    call("nnz_ss2d_xproj_backward_w_ws", ptr(dP), ptr(x2), ptr(d_xproj), ptr(ws), nws, B, Di, 2 * Cp, L, Cp,
                         stream_ptr())
                else:
                    d_xproj = torch.zeros((K, Cp, Di), **f32)
                    call("nnz_ss2d_xproj_backward_w", ptr(dP), ptr(x2), ptr(d_xproj), B, Di, 2 * Cp, L, Cp, stream_ptr())
            else:
                dWst = _proj_weight_grad(dP, x2)                                 # (2, 2Cp, Di)
                d_xproj = dWst.view(2, 2, Cp, Di).transpose(0, 1).reshape(K, Cp, Di)
        return dx2, None, d_xproj, dWdt.view(K, Di, R), dbias.view(K, Di), dA, dD    # dA is dA_log (a_is_log)


def ss2d_cross_scan(xc, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
    """xc: (B, Di, H, W) conv+SiLU output (fp16 or fp32) -> (B, H, W, Di) fp32: the sum of the four directional scans
    in token-major layout (what SS2D.forward feeds to out_norm)."""
    if not xc.is_cuda:
        raise RuntimeError("ss2d_cross_scan runs on MI355X through libnnuzoo_hip.so only (no CPU fallback)")
    H, W = xc.shape[2:]
    return _SS2DCrossScan.apply(_PrepareFn.apply(xc), (H, W), x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)


def dwconv_supported(conv: torch.nn.Conv2d) -> bool:
    return conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.stride == (1, 1) and conv.dilation == (1, 1) \
        and conv.groups == conv.in_channels == conv.out_channels and conv.padding_mode == "zeros"


def ss2d_conv_cross_scan(x_tok, conv, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
    """x_tok: (B, H, W, Di) token-major x half of the in_proj output (a strided view is read in place); applies the block's
    depthwise 3x3 conv + SiLU and the four-direction scan -> (B, H, W, Di) fp32."""
    if not x_tok.is_cuda:
        raise RuntimeError("ss2d_conv_cross_scan runs on MI355X through libnnuzoo_hip.so only (no CPU fallback)")
    H, W = x_tok.shape[1:3]
    x2 = _DwConvSiluPrepareFn.apply(x_tok, conv.weight, conv.bias)
    return _SS2DCrossScan.apply(x2, (H, W), x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)
