"""`LayerNorm` with nn.LayerNorm's constructor, parameters and state_dict, running on the hand-written gfx950 kernels
(csrc/layer_norm.hip) - the normalisation of the VSS / SSND / Swin blocks (reference: nn.LayerNorm at m2net.py:101,521,
ssnd2net.py:266,532, swt2net.py:630-660).  Numerics follow torch under autocast: statistics and output in fp32 whatever
the input type (layer_norm is on autocast's fp32 list), gradient of the input in the input's type.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from ._lib import call, ptr, stream_ptr


def _dy(dy: torch.Tensor) -> torch.Tensor:
    """the kernels read f32 or f16 gradients in place"""
    return (dy if dy.dtype in (torch.float16, torch.float32) else dy.float()).contiguous()


# NNZ_LN_DET=0: dgamma / dbeta through fp32 atomics (rounds 1-2) instead of the fixed-point sums + last-workgroup finalisation
_LN_DET = os.environ.get("NNZ_LN_DET", "1") != "0"


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        C = x.shape[-1]
        xc = x.contiguous()
        rows = xc.numel() // C
        half = out_dtype == torch.float16
        y = torch.empty(xc.shape, dtype=torch.float16 if half else torch.float32, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        # dgamma / dbeta accumulator of the backward, zeroed by the forward launch (no zeroing launch later)
        gwb = None     # round 3: the deterministic backward WRITES dgamma / dbeta, nothing to pre-zero
        call("nnz_layer_norm_forward", ptr(xc), int(xc.dtype == torch.float16), ptr(weight), ptr(bias), ptr(y), int(half),
             ptr(mean), ptr(rstd), ptr(gwb), rows, C, float(eps), stream_ptr())
        ctx.save_for_backward(xc, weight, mean, rstd)
        ctx.gwb = gwb
        ctx.has_bias = bias is not None
        return y if out_dtype in (torch.float32, torch.float16) else y.to(out_dtype)

    @staticmethod
    def backward(ctx, dy):
        xc, weight, mean, rstd = ctx.saved_tensors
        C = xc.shape[-1]
        rows = xc.numel() // C
        dy = _dy(dy)
        dx = torch.empty_like(xc)
        dw, db, pre = _affine_grads(weight, ctx.has_bias, ctx.needs_input_grad[1], ctx.needs_input_grad[2], C, xc.device,
                                    ctx.gwb)
        ctx.gwb = None      # single use: a second backward through the same node zeroes in its own launch
        if _LN_DET:
            from .hip_ops import det_scratch
            sc = det_scratch(xc.device, 2 * C)     # fixed-point cross-workgroup sums: dgamma / dbeta bit-identical run to run
            call("nnz_layer_norm_backward_det", ptr(xc), int(xc.dtype == torch.float16), ptr(weight), ptr(mean), ptr(rstd),
                 ptr(dy), int(dy.dtype == torch.float16), ptr(dx), ptr(dw), ptr(db), ptr(sc.acc), ptr(sc.counter), rows, C,
                 stream_ptr())
        else:
            call("nnz_layer_norm_backward", ptr(xc), int(xc.dtype == torch.float16), ptr(weight), ptr(mean), ptr(rstd),
                 ptr(dy), int(dy.dtype == torch.float16), ptr(dx), ptr(dw), ptr(db), int(pre), rows, C, stream_ptr())
        return dx, dw, db, None, None


class _LayerNormSkipFn(torch.autograd.Function):
    """x -> (LayerNorm(x), x) for the pre-norm residual blocks (x + f(norm(x)): swt2net.py:646-659, m2net.py:530).  The second
    output is x itself; taking the residual stream from it instead of from the caller's x brings both gradients of x into ONE
    backward call, where the kernel adds them (nnz_layer_norm_backward_det_res) - autograd's accumulation is an add launch per
    norm, 288 per SwT2Net step, 80 per M2Net step.  fp32 or fp16 contiguous device tensors; `half_out`: fp16 rows for an autocast
    Linear (LayerNorm.feeds_linear)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, half_out):
        C = x.shape[-1]
        rows = x.numel() // C
        xh = int(x.dtype == torch.float16)
        y = torch.empty(x.shape, dtype=torch.float16 if half_out else torch.float32, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        call("nnz_layer_norm_forward", ptr(x), xh, ptr(weight), ptr(bias), ptr(y), int(half_out), ptr(mean), ptr(rstd), None, rows,
             C, float(eps), stream_ptr())
        ctx.save_for_backward(x, weight, mean, rstd)
        ctx.has_bias = bias is not None
        ctx.set_materialize_grads(False)
        return y, x

    @staticmethod
    def backward(ctx, dy, dskip):
        if dy is None:
            return dskip, None, None, None, None
        x, weight, mean, rstd = ctx.saved_tensors
        C = x.shape[-1]
        rows = x.numel() // C
        dy = _dy(dy)
        if dskip is not None and (dskip.dtype != x.dtype or not dskip.is_contiguous()):
            dskip = dskip.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        dw, db, _ = _affine_grads(weight, ctx.has_bias, ctx.needs_input_grad[1], ctx.needs_input_grad[2], C, x.device)
        from .hip_ops import det_scratch
        sc = det_scratch(x.device, 2 * C)
        call("nnz_layer_norm_backward_det_res", ptr(x), int(x.dtype == torch.float16), ptr(weight), ptr(mean), ptr(rstd), ptr(dy),
             int(dy.dtype == torch.float16), ptr(dskip), ptr(dx), ptr(dw), ptr(db), ptr(sc.acc), ptr(sc.counter), rows, C,
             stream_ptr())
        return dx, dw, db, None, None


def layer_norm_skip(norm: "nn.LayerNorm", x: torch.Tensor):
    """(norm(x), x) with the two gradients of x summed inside the LayerNorm backward kernel; falls back to (norm(x), x) when
    the tensor is not a contiguous fp32 / fp16 device tensor the kernel takes.  Output type as `layer_norm`: fp32, or fp16 rows
    under fp16 autocast when the norm feeds an autocast Linear (`feeds_linear`)."""
    C = x.shape[-1]
    ac = torch.is_autocast_enabled()
    ok = isinstance(norm, nn.LayerNorm) and norm.elementwise_affine and x.is_cuda and x.is_contiguous() \
        and (x.dtype == torch.float32 or (x.dtype == torch.float16 and ac)) \
        and len(norm.normalized_shape) == 1 and norm.normalized_shape[0] == C \
        and C % 4 == 0 and C <= 2048 and torch.is_grad_enabled() and x.requires_grad \
        and (norm.weight is None or norm.weight.dtype == torch.float32) and _LN_DET
    if not ok:
        return norm(x), x
    return _LayerNormSkipFn.apply(x, norm.weight, norm.bias, norm.eps, _half_for_linear(getattr(norm, "feeds_linear", False)))


def _affine_grads(weight, has_bias, need_w, need_b, C, device, prezeroed=None):
    """dgamma / dbeta as the two rows of one buffer when both are needed (zeroed by the forward launch when `prezeroed`
    is that buffer, else by one launch of the backward).  Returns (dgamma, dbeta, pre_zeroed flag)."""
    need_w, need_b = need_w and weight is not None, need_b and has_bias
    if need_w and need_b:
        if prezeroed is not None:
            return prezeroed[0], prezeroed[1], 1
        both = torch.empty((2, C), dtype=torch.float32, device=device)
        return both[0], both[1], 0
    return (torch.empty(C, dtype=torch.float32, device=device) if need_w else None,
            torch.empty(C, dtype=torch.float32, device=device) if need_b else None, 0)


def _row_stride(z: torch.Tensor):
    """element stride between consecutive rows of z viewed as [rows][C], or None when z is not such a view"""
    C = z.shape[-1]
    if z.stride(-1) != 1:
        return None
    rs = z.stride(-2) if z.dim() > 1 else C
    expect = rs
    for d in range(z.dim() - 2, -1, -1):
        if z.shape[d] != 1 and z.stride(d) != expect:
            return None
        expect *= z.shape[d]
    return rs


class _LayerNormGateFn(torch.autograd.Function):
    """LayerNorm(x) * silu(z) (SS2D's gated output norm) in one pass each way"""

    @staticmethod
    def forward(ctx, x, z, weight, bias, eps, half_out):
        C = x.shape[-1]
        xc = x.contiguous()
        rows = xc.numel() // C
        zs = _row_stride(z)
        if zs is None or zs % 4 or z.data_ptr() % 8:
            z = z.contiguous()
            zs = C
        y = torch.empty(xc.shape, dtype=torch.float16 if half_out else torch.float32, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        gwb = None
        call("nnz_layer_norm_gate_forward", ptr(xc), int(xc.dtype == torch.float16), ptr(weight), ptr(bias), ptr(z),
             int(z.dtype == torch.float16), zs, ptr(y), int(half_out), ptr(mean), ptr(rstd), ptr(gwb), rows, C,
             float(eps), stream_ptr())
        ctx.save_for_backward(xc, z, weight, bias, mean, rstd)
        ctx.zs, ctx.gwb = zs, gwb
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, z, weight, bias, mean, rstd = ctx.saved_tensors
        C = xc.shape[-1]
        rows = xc.numel() // C
        dy = _dy(dy)
        dx = torch.empty_like(xc)
        dz = torch.empty(z.shape, dtype=z.dtype, device=z.device)
        dw, db, pre = _affine_grads(weight, bias is not None, ctx.needs_input_grad[2], ctx.needs_input_grad[3], C,
                                    xc.device, ctx.gwb)
        ctx.gwb = None
        if _LN_DET:
            from .hip_ops import det_scratch
            sc = det_scratch(xc.device, 2 * C)
            call("nnz_layer_norm_gate_backward_det", ptr(xc), int(xc.dtype == torch.float16), ptr(weight), ptr(bias), ptr(z),
                 int(z.dtype == torch.float16), ctx.zs, ptr(mean), ptr(rstd), ptr(dy), int(dy.dtype == torch.float16),
                 ptr(dx), ptr(dz), ptr(dw), ptr(db), ptr(sc.acc), ptr(sc.counter), rows, C, stream_ptr())
        else:
            call("nnz_layer_norm_gate_backward", ptr(xc), int(xc.dtype == torch.float16), ptr(weight), ptr(bias), ptr(z),
                 int(z.dtype == torch.float16), ctx.zs, ptr(mean), ptr(rstd), ptr(dy), int(dy.dtype == torch.float16),
                 ptr(dx), ptr(dz), ptr(dw), ptr(db), int(pre), rows, C, stream_ptr())
        return dx, dz, dw, db, None, None


def _half_for_linear(flag: bool) -> bool:
    """write fp16 when the only consumer is a Linear under fp16 autocast: that Linear would round the same fp32 value to
    fp16 itself, so the numbers are identical and a cast pass (and its backward) disappears"""
    return bool(flag) and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.float16


def layer_norm_gate(x: torch.Tensor, z: torch.Tensor, weight, bias, eps: float = 1e-5,
                    feeds_linear: bool = False) -> torch.Tensor:
    """F.layer_norm(x, (C,), weight, bias, eps) * F.silu(z) -> fp32 (fp16 under autocast when `feeds_linear`), one kernel
    each way (no CPU path)"""
    if not x.is_cuda:
        raise RuntimeError("nnuzoo_amd.layer_norm runs on MI355X through libnnuzoo_hip.so only (no CPU fallback)")
    C = x.shape[-1]
    if C % 4 or C > 2048 or x.dtype not in (torch.float16, torch.float32) or z.shape != x.shape or \
            z.dtype not in (torch.float16, torch.float32) or (weight is not None and weight.dtype != torch.float32):
        raise NotImplementedError(f"layer_norm_gate kernel: C % 4 == 0, C <= 2048, fp16/fp32 x and z of one shape "
                                  f"(got C={C}, {x.dtype}, {z.dtype})")
    return _LayerNormGateFn.apply(x, z, weight, bias, eps, _half_for_linear(feeds_linear))


def layer_norm(x: torch.Tensor, weight, bias, eps: float = 1e-5, feeds_linear: bool = False) -> torch.Tensor:
    """F.layer_norm(x, (C,), weight, bias, eps) over the last dimension on the HIP kernel (no CPU path)."""
    if not x.is_cuda:
        raise RuntimeError("nnuzoo_amd.layer_norm runs on MI355X through libnnuzoo_hip.so only (no CPU fallback)")
    C = x.shape[-1]
    if C % 4 or C > 2048 or x.dtype not in (torch.float16, torch.float32) or \
            (weight is not None and weight.dtype != torch.float32):
        raise NotImplementedError(f"layer_norm kernel: C % 4 == 0, C <= 2048, fp16/fp32 input, fp32 affine "
                                  f"(got C={C}, {x.dtype})")
    # torch semantics: fp32 result under autocast (layer_norm is on the fp32 list), else the input's type
    out_dtype = torch.float32 if (torch.is_autocast_enabled() or x.dtype == torch.float32) else x.dtype
    if _half_for_linear(feeds_linear):
        out_dtype = torch.float16
    return _LayerNormFn.apply(x, weight, bias, eps, out_dtype)


class LayerNorm(nn.LayerNorm):
    """drop-in for nn.LayerNorm (normalised shape = the last dimension).  `feeds_linear` (set by the owning block when the
    output goes into an nn.Linear and nowhere else) lets the kernel write fp16 under fp16 autocast."""
    feeds_linear = False

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if len(self.normalized_shape) != 1:
            raise NotImplementedError("nnuzoo_amd.LayerNorm normalises the last dimension only")
        return layer_norm(x, self.weight, self.bias, self.eps, self.feeds_linear)
