"""`selective_scan_fn` with the call signature of mamba_ssm.ops.selective_scan_interface.selective_scan_fn, backed by
the hand-written gfx950 chunk-scan kernels (csrc/selective_scan.hip).  This is the function the reference's SS2D /
SSND blocks bind as `self.selective_scan` (/root/reference/nnunetv2/nets/m2net.py:11,107,193-199;
ssnd2net.py:18,176,271-277); the in-tree autograd wrapper it mirrors is
/root/reference/nnunetv2/nets/seg_mamba/selective_scan_interface.py:14-83.

Supported (exactly what those call sites use): real A of shape (K*D, 16), B and C of shape (B, K, 16, L) (or
(B, 16, L) = one group), fp32, optional z gate (1-D Mamba block).  Anything else raises - there is no eager fallback.
"""
from __future__ import annotations

import torch

from ._lib import call, load, ptr, stream_ptr


def _prep(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"selective_scan_fn: `{name}` is a CPU tensor; the scan runs on MI355X through "
                           "libnnuzoo_hip.so only (no CPU fallback; oracle/selective_scan.py is test-only)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class SelectiveScanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                return_last_state=False):
        if z is not None:
            raise NotImplementedError("SelectiveScanFn: z gating is applied by selective_scan_fn (separate kernel)")
        if return_last_state:
            raise NotImplementedError("selective_scan_fn: return_last_state is not used by the SS2D/SSND call sites")
        if A.is_complex():
            raise NotImplementedError("selective_scan_fn: complex A is not supported")
        u, delta, A, B, C = _prep(u, "u"), _prep(delta, "delta"), _prep(A, "A"), _prep(B, "B"), _prep(C, "C")
        D, delta_bias = _prep(D, "D"), _prep(delta_bias, "delta_bias")
        squeeze = False
        if B.dim() == 3:
            B, C, squeeze = B.unsqueeze(1), C.unsqueeze(1), True
        Bt, KD, L = u.shape
        K, N = B.shape[1], A.shape[1]
        if N != 16 or KD % K or B.shape != (Bt, K, N, L) or C.shape != B.shape or A.shape[0] != KD:
            raise NotImplementedError(f"selective_scan_fn: unsupported shapes u{tuple(u.shape)} A{tuple(A.shape)} "
                                      f"B{tuple(B.shape)} (need d_state 16, B/C (b, K, 16, L))")
        Dg = KD // K
        lib = load()
        dev = u.device
        y = torch.empty_like(u)
        state = torch.empty(lib.nnz_selective_scan_state_floats(Bt, KD, L), dtype=torch.float32, device=dev)
        ws = torch.empty(lib.nnz_selective_scan_workspace_floats(Bt, KD, L), dtype=torch.float32, device=dev)
        call("nnz_selective_scan_forward", ptr(u), ptr(delta), ptr(A), ptr(B), ptr(C), ptr(D), ptr(delta_bias),
             ptr(y), ptr(state), ptr(ws), Bt, K, Dg, N, L, int(delta_softplus), stream_ptr())
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias, state)
        ctx.cfg = (Bt, K, Dg, N, L, int(delta_softplus), squeeze)
        return y

    @staticmethod
    def backward(ctx, dy):
        u, delta, A, B, C, D, delta_bias, state = ctx.saved_tensors
        Bt, K, Dg, N, L, sp, squeeze = ctx.cfg
        lib = load()
        dev = u.device
        dy = dy.float().contiguous()
        KD = K * Dg
        du, ddelta = torch.empty_like(u), torch.empty_like(delta)
        dA = torch.empty_like(A)
        dB, dC = torch.empty_like(B), torch.empty_like(C)
        dD = torch.empty_like(D) if D is not None else None
        dbias = torch.empty_like(delta_bias) if delta_bias is not None else None
        gstate = torch.empty(lib.nnz_selective_scan_grad_state_floats(Bt, KD, L), dtype=torch.float32, device=dev)
        ws = torch.empty(lib.nnz_selective_scan_workspace_floats(Bt, KD, L), dtype=torch.float32, device=dev)
        call("nnz_selective_scan_backward", ptr(u), ptr(delta), ptr(A), ptr(B), ptr(C), ptr(D), ptr(delta_bias),
             ptr(dy), ptr(state), ptr(gstate), ptr(ws), ptr(du), ptr(ddelta), ptr(dA), ptr(dB), ptr(dC), ptr(dD),
             ptr(dbias), Bt, K, Dg, N, L, sp, stream_ptr())
        if squeeze:
            dB, dC = dB.squeeze(1), dC.squeeze(1)
        return du, ddelta, dA, dB, dC, dD, None, dbias, None, None


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    y = SelectiveScanFn.apply(u, delta, A, B, C, D, None, delta_bias, delta_softplus, return_last_state)
    if z is not None:
        # the Mamba block's gate: out = y * silu(z) (selective_scan_ref, selective_scan_interface.py:140-148)
        from .mamba_block import silu_gate
        y = silu_gate(y, z)
    return y
