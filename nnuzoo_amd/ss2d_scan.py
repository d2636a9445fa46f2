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

import torch

from ._lib import call, load, ptr, stream_ptr

N_STATE = 16
MAX_DT_RANK = 8


def supported(x: torch.Tensor, dt_rank: int, d_state: int) -> bool:
    return x.is_cuda and x.dim() == 4 and d_state == N_STATE and 1 <= dt_rank <= MAX_DT_RANK and x.shape[1] % 4 == 0 \
        and x.dtype in (torch.float16, torch.float32) and x.shape[0] * x.shape[1] <= 65535


class _SS2DCrossScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xc, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
        lib = load()
        B, Di, H, W = xc.shape
        L, K = H * W, 4
        R = dt_projs_weight.shape[2]
        Cp = R + 2 * N_STATE
        dev = xc.device
        f32 = dict(dtype=torch.float32, device=dev)
        xc = xc.contiguous()
        with torch.autocast("cuda", enabled=False):
            x2 = torch.empty((2, B, Di, L), **f32)
            call("nnz_ss2d_prepare", ptr(xc), int(xc.dtype == torch.float16), ptr(x2), B, Di, H, W, stream_ptr())
            # direction k = s + 2j  ->  rows [j*Cp, (j+1)*Cp) of source s's stacked projection weight
            Wst = x_proj_weight.detach().float().view(2, 2, Cp, Di).transpose(0, 1).reshape(2, 1, 2 * Cp, Di)
            P = torch.matmul(Wst, x2)                                            # (2, B, 2Cp, L)
            A = A_logs.detach().float().contiguous()                             # A_log; the kernels use -exp(A_log)
            Wdt = dt_projs_weight.detach().float().reshape(K * Di, R).contiguous()
            bias = dt_projs_bias.detach().float().reshape(-1).contiguous()
            Dv = Ds.detach().float().contiguous()
            y = torch.empty((B, K, Di, L), **f32)
            state = torch.empty(lib.nnz_selective_scan_state_floats(B, K * Di, L), **f32)
            ws = torch.empty(lib.nnz_selective_scan_workspace_floats(B, K * Di, L), **f32)
            call("nnz_ss2d_scan_forward", ptr(x2), ptr(P), ptr(Wdt), ptr(A), ptr(Dv), ptr(bias), ptr(y), ptr(state),
                 ptr(ws), B, Di, R, L, 1, 1, stream_ptr())
            out = torch.empty((B, H, W, Di), **f32)
            call("nnz_ss2d_merge", ptr(y), ptr(out), B, Di, H, W, stream_ptr())
        ctx.save_for_backward(x2, P, Wst, A, Wdt, bias, Dv, state)
        ctx.meta = (B, Di, H, W, R, xc.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = load()
        x2, P, Wst, A, Wdt, bias, Dv, state = ctx.saved_tensors
        B, Di, H, W, R, xdtype = ctx.meta
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
            gstate = torch.empty_like(state)
            ws = torch.empty(lib.nnz_selective_scan_workspace_floats(B, K * Di, L), **f32)
            call("nnz_ss2d_scan_backward", ptr(x2), ptr(P), ptr(Wdt), ptr(A), ptr(Dv), ptr(bias), ptr(dy2), ptr(state),
                 ptr(gstate), ptr(ws), ptr(du), ptr(dP), ptr(dWdt), ptr(dA), ptr(dD), ptr(dbias), B, Di, R, L, 1, 1,
                 stream_ptr())
            dx2 = torch.matmul(Wst.transpose(-1, -2), dP)                        # (2, B, Di, L)
            dWst = torch.einsum("sbcl,sbdl->scd", dP, x2)                        # (2, 2Cp, Di)
            dx = torch.empty((B, Di, H, W), dtype=xdtype, device=dout.device)
            call("nnz_ss2d_merge_dx", ptr(du), ptr(dx2), ptr(dx), int(xdtype == torch.float16), B, Di, H, W,
                 stream_ptr())
            d_xproj = dWst.view(2, 2, Cp, Di).transpose(0, 1).reshape(K, Cp, Di)
        return dx, d_xproj, dWdt.view(K, Di, R), dbias.view(K, Di), dA, dD    # dA is dA_log (a_is_log)


def ss2d_cross_scan(xc, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
    """xc: (B, Di, H, W) conv+SiLU output (fp16 or fp32) -> (B, H, W, Di) fp32: the sum of the four directional scans
    in token-major layout (what SS2D.forward feeds to out_norm)."""
    if not xc.is_cuda:
        raise RuntimeError("ss2d_cross_scan runs on MI355X through libnnuzoo_hip.so only (no CPU fallback)")
    return _SS2DCrossScan.apply(xc, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)
