"""SwT2Net's RSU4F stages in fp32 on hand-written kernels, channels-last (round 6).

Reference: /root/reference/nnunetv2/nets/swt2net.py:17-31 REBNCONV = get_dwconv_layer (depthwise 3x3 + pointwise 1x1, both bias-free)
-> nn.BatchNorm2d -> ReLU; RSU4F :873-905 = eight of them with channel concatenations and a residual sum; the trainer's step is fp32
without autocast (nnUNetTrainerSwT2Net.py:112-130).  Through torch these are MIOpen convolutions (solver chosen per process), NCHW <->
NHWC transposes, MIOpen batch norm and ATen's depthwise kernels: ~140 launches and 2.5 ms of the 56 ms SwT2Net step, and the reason
the step time differed by 7 % from box to box (DESIGN of round 5, section 5).

Here an RSU4F converts to token-major [B, H, W, C] once (free when the input is the permuted view the Swin stages hand over) and
every unit is: depthwise 3x3 (csrc/sepconv32.hip) -> the pointwise 1x1 as a token Linear on the fp32 MFMA kernels
(csrc/dense32.hip; its weight gradient rides in the pass's grouped launch) -> BatchNorm with batch statistics + ReLU in one launch
(csrc/sepconv32.hip).  The concatenations are last-dimension `torch.cat`s.  All reductions run in a fixed order: the stage is
bit-reproducible.  Eval-mode statistics WITH autograd (frozen-BN fine-tuning) keep the torch path, like nnuzoo_amd/rebnconv.py."""
from __future__ import annotations

import os

import torch

from . import _lib
from ._lib import call, ptr, stream_ptr

USE_HIP = os.environ.get("NNZ_SEPCONV32", "1") != "0"      # A/B switch for measurements
SMALL_WGRAD_MIN_TOKENS = int(os.environ.get("NNZ_PW_SMALL_WGRAD_TOKENS", "16384"))


class _Dw3x3Fn(torch.autograd.Function):
    """x [B, H, W, C] fp32 contiguous, weight [C, 1, 3, 3] (a leaf parameter), bias [C] or None"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, H, W, C = x.shape
        y = torch.empty_like(x)
        call("nnz_dw3x3_nhwc_f32", ptr(x), ptr(weight), ptr(bias), ptr(y), B, H, W, C, 0, stream_ptr())
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B, H, W, C = x.shape
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            call("nnz_dw3x3_nhwc_f32", ptr(dy), ptr(weight), None, ptr(dx), B, H, W, C, 1, stream_ptr())
        if ctx.needs_input_grad[1]:
            ws = torch.empty(int(_lib.load().nnz_dw3x3_nhwc_wgrad_workspace_floats(B, H, W, C)), dtype=torch.float32, device=x.device)
            dw = torch.empty_like(weight)
            call("nnz_dw3x3_nhwc_wgrad_f32", ptr(x), ptr(dy), ptr(ws), ptr(dw), B, H, W, C, stream_ptr())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 1, 2))
        return dx, dw, db


class _Conv1x1Param:
    """what token_linear's grouped weight-gradient launch needs of a Linear weight - a 2-D shape and a `.grad` slot - for the
    [N, K, 1, 1] parameter of a pointwise convolution (same memory: the launch writes [N][K] floats)"""

    def __init__(self, p: torch.nn.Parameter):
        self.p = p
        self.shape = (p.shape[0], p[0].numel())

    @property
    def grad(self):
        g = self.p.grad
        return None if g is None else g.view(self.shape)

    @grad.setter
    def grad(self, g):
        self.p.grad = g.view(self.p.shape)


class _Pointwise1x1Fn(torch.autograd.Function):
    """y[t] = W x[t] (+ b) over the tokens of x [B, H, W, K]; W is a conv parameter [N, ...] read as the [N, K] matrix it is in memory:
    [N, K, 1, 1] of a pointwise convolution, or [N, C, p, p] of a kernel = stride patch embedding applied to space-to-depth tokens
    (K = C p p)"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from .token_linear import _d32_forward
        N, K = weight.shape[0], weight[0].numel()
        x2 = x.reshape(-1, K)
        T = x2.shape[0]
        y = torch.empty((T, N), dtype=x2.dtype, device=x.device)      # fp32, or fp16 rows in / fp16 rows out (autocast nets)
        _d32_forward(x2, weight, bias, y, None, T, K, N, 0)
        ctx.save_for_backward(x2, weight)
        ctx.params = (weight, bias)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        from .token_linear import _DEFER, _d32_dgrad, _d32_workspace, _has_grad_hooks
        x2, weight = ctx.saved_tensors
        wp, bp = ctx.params
        N, K = weight.shape[0], weight[0].numel()
        T = x2.shape[0]
        dy2 = dy.reshape(-1, N)
        if not dy2.is_contiguous() or dy2.dtype != x2.dtype:
            dy2 = dy2.to(x2.dtype).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((T, K), dtype=x2.dtype, device=dy.device)
            _d32_dgrad(dy2, weight.view(N, K), None, dx, T, K, N)
            dx = dx.view(ctx.xshape)
        need_w, need_b = ctx.needs_input_grad[1], bp is not None and ctx.needs_input_grad[2]
        if need_w and N <= 64 and K <= 64 and T >= SMALL_WGRAD_MIN_TOKENS:
            # few channels over very many tokens (stems / heads / 1x1 embeddings of the full-resolution stages): a streaming reduction on
            # its own kernel (csrc/sepconv32.hip pw_wgrad_small_kernel; fp32 or fp16 rows, the bias gradient from the same pass) -
            # half-empty 64-wide MFMA tiles cost ~140 us per problem in the grouped launch
            lib = _lib.load()
            ws = torch.empty(int(lib.nnz_pw_wgrad_small_workspace_floats_b(T, N, K, int(need_b))), dtype=torch.float32, device=dy.device)
            dw = torch.empty_like(weight)
            db = torch.empty(N, dtype=torch.float32, device=dy.device) if need_b else None
            call("nnz_pw_wgrad_small", ptr(dy2), ptr(x2), int(x2.dtype == torch.float16), ptr(ws), ptr(dw), ptr(db), T, N, K, stream_ptr())
            return dx, dw, db
        if need_w or need_b:
            if _DEFER["on"] and need_w and wp.is_leaf and (bp is None or bp.is_leaf) and not _has_grad_hooks(wp) \
                    and not (bp is not None and _has_grad_hooks(bp)):
                _DEFER["jobs"].append((dy2, x2, _Conv1x1Param(wp), bp if need_b else None))
                return dx, None, None
            if dy2.dtype == torch.float16:     # outside a deferred pass (module-level tests): the fp32 entry point on fp32 copies
                dy2, x2 = dy2.float(), x2.float()
            dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
            db = torch.empty(N, dtype=torch.float32, device=dy.device) if need_b else None
            ws = _d32_workspace(dy.device, int(_lib.load().nnz_dense32_wgrad_workspace_floats(T, K, N)))
            call("nnz_dense32_wgrad", ptr(dy2), ptr(x2), ptr(dw), ptr(db), ptr(ws), T, K, N, stream_ptr())
            dw = dw.view(weight.shape)
        return dx, dw, db


class _BnReluFn(torch.autograd.Function):
    """relu(batch_norm(x)) on [..., C] fp32: batch statistics when `training`, the running estimates otherwise"""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, training):
        C = x.shape[-1]
        T = x.numel() // C
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty(C, dtype=torch.float32, device=x.device)
        call("nnz_bn_relu_nhwc_forward_f32", ptr(x), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(mean), ptr(rstd),
             ptr(y), T, C, int(training), float(momentum), float(eps), stream_ptr())
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.training = bool(training)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("nnuzoo_amd sepconv32: backward through eval-mode BatchNorm statistics is not supported")
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        C = x.shape[-1]
        T = x.numel() // C
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgb = torch.empty((2, C), dtype=torch.float32, device=x.device)
        call("nnz_bn_relu_nhwc_backward_f32", ptr(x), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dx), ptr(dgb[0]),
             ptr(dgb[1]), T, C, stream_ptr())
        return dx, dgb[0], dgb[1], None, None, None, None, None


def _parts(unit):
    """(depthwise conv, pointwise conv, batch norm) of a swt2net REBNCONV, or None when the unit is something else"""
    seq = getattr(unit, "conv_s1", None)
    bn = getattr(unit, "bn_s1", None)
    if not isinstance(seq, torch.nn.Sequential) or len(seq) != 2 or type(bn) is not torch.nn.BatchNorm2d:
        return None
    dw, pw = getattr(seq[0], "conv", None), getattr(seq[1], "conv", None)
    if not isinstance(dw, torch.nn.Conv2d) or not isinstance(pw, torch.nn.Conv2d):
        return None
    ok = dw.groups == dw.in_channels == dw.out_channels and dw.kernel_size == (3, 3) and dw.stride == (1, 1) \
        and dw.padding == (1, 1) and dw.dilation == (1, 1) and dw.padding_mode == "zeros" and dw.in_channels % 4 == 0 \
        and pw.kernel_size == (1, 1) and pw.stride == (1, 1) and pw.padding == (0, 0) and pw.groups == 1 \
        and pw.in_channels % 4 == 0 and pw.out_channels % 4 == 0 \
        and bn.affine and bn.track_running_stats and bn.momentum is not None
    return (dw, pw, bn) if ok else None


def hip_path_ok(rsu, x: torch.Tensor) -> bool:
    if not (USE_HIP and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and not torch.is_autocast_enabled()):
        return False
    units = [rsu.rebnconvin, rsu.rebnconv1, rsu.rebnconv2, rsu.rebnconv3, rsu.rebnconv4, rsu.rebnconv3d, rsu.rebnconv2d,
             rsu.rebnconv1d]
    parts = [_parts(u) for u in units]
    if any(p is None for p in parts):
        return False
    if any(p.dtype != torch.float32 for u in units for p in u.parameters()):
        return False
    if x.shape[0] * x.shape[2] * x.shape[3] < 64:          # dense32's smallest token count (token_linear.DENSE32_MIN_TOKENS)
        return False
    bn0 = parts[0][2]
    if not bn0.training and torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in rsu.parameters())):
        return False      # eval-mode statistics with autograd: torch path
    return True


def unit_nhwc(unit, x: torch.Tensor) -> torch.Tensor:
    dw, pw, bn = _parts(unit)
    if bn.training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    h = _Dw3x3Fn.apply(x, dw.weight, dw.bias)
    h = _Pointwise1x1Fn.apply(h, pw.weight, pw.bias)
    return _BnReluFn.apply(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, bn.training)


def rsu4f_forward(rsu, x: torch.Tensor) -> torch.Tensor:
    """RSU4F.forward (swt2net.py:889-905) on token-major fp32 activations; input / output NCHW like the module"""
    xc = x.permute(0, 2, 3, 1).contiguous()
    xin = unit_nhwc(rsu.rebnconvin, xc)
    e1 = unit_nhwc(rsu.rebnconv1, xin)
    e2 = unit_nhwc(rsu.rebnconv2, e1)
    e3 = unit_nhwc(rsu.rebnconv3, e2)
    e4 = unit_nhwc(rsu.rebnconv4, e3)
    d3 = unit_nhwc(rsu.rebnconv3d, torch.cat((e4, e3), -1))
    d2 = unit_nhwc(rsu.rebnconv2d, torch.cat((d3, e2), -1))
    d1 = unit_nhwc(rsu.rebnconv1d, torch.cat((d2, e1), -1))
    return (d1 + xin).permute(0, 3, 1, 2)


def _fp32_device(x: torch.Tensor) -> bool:
    return USE_HIP and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()


def _h16_autocast(x: torch.Tensor) -> bool:
    """a device tensor inside an fp16-autocast region (the step of the Mamba / U^2 nets): the kernels read fp16 activations (an fp32
    tensor is rounded to fp16 first - what autocast does in front of its convolution) and the fp32 master weights, and write
    autocast's output type"""
    return USE_HIP and x.is_cuda and x.dtype in (torch.float16, torch.float32) and torch.is_autocast_enabled() \
        and torch.get_autocast_dtype("cuda") == torch.float16


def _h16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float16) if x.dtype == torch.float32 and torch.is_autocast_enabled() else x


def stem_ok(seq, x: torch.Tensor) -> bool:
    """get_dwconv_layer (depthwise 3x3 + pointwise 1x1, no norm) of a Swin U-net stage on a fp32 NCHW device tensor"""
    if not (_fp32_device(x) and x.dim() == 4 and isinstance(seq, torch.nn.Sequential) and len(seq) == 2):
        return False
    dw, pw = getattr(seq[0], "conv", None), getattr(seq[1], "conv", None)
    if not isinstance(dw, torch.nn.Conv2d) or not isinstance(pw, torch.nn.Conv2d):
        return False
    return dw.groups == dw.in_channels == dw.out_channels and dw.kernel_size == (3, 3) and dw.stride == (1, 1) \
        and dw.padding == (1, 1) and dw.dilation == (1, 1) and dw.padding_mode == "zeros" \
        and (dw.in_channels % 4 == 0 or dw.in_channels < 4) \
        and pw.kernel_size == (1, 1) and pw.stride == (1, 1) and pw.padding == (0, 0) and pw.groups == 1 \
        and pw.out_channels % 4 == 0 and dw.weight.dtype == torch.float32 and pw.weight.dtype == torch.float32 \
        and x.shape[0] * x.shape[2] * x.shape[3] >= 64


def stem_forward(seq, x: torch.Tensor) -> torch.Tensor:
    """NCHW in, NCHW view of token-major storage out.  A stem with fewer than four channels (the 1-channel network input of stage 1)
    runs with its channels zero-padded to four: the image and the two small weights are padded on the fly (the extra channels
    contribute exact zeros; autograd slices the gradients back), so the same kernels serve it"""
    dw, pw = seq[0].conv, seq[1].conv
    xt = x.permute(0, 2, 3, 1)
    wd, bd, wp = dw.weight, dw.bias, pw.weight
    C = dw.in_channels
    if C % 4:
        pad = 4 - C % 4
        xt = torch.nn.functional.pad(xt, (0, pad))
        wd = torch.nn.functional.pad(wd, (0, 0, 0, 0, 0, 0, 0, pad))
        bd = torch.nn.functional.pad(bd, (0, pad)) if bd is not None else None
        wp = torch.nn.functional.pad(wp, (0, 0, 0, 0, 0, pad))
    h = _Dw3x3Fn.apply(xt.contiguous(), wd, bd)
    return _Pointwise1x1Fn.apply(h, wp, pw.bias).permute(0, 3, 1, 2)


def pointwise_ok(conv, x_tokens: torch.Tensor) -> bool:
    """a 1x1 nn.Conv2d applied to token-major activations [B, H, W, K]: fp32 of a fp32 device step, or fp16 inside an fp16-autocast
    region (fp16 rows in and out, the fp32 master weight as it is: csrc/dense32.hip *_h16)"""
    return (_fp32_device(x_tokens) or _h16_autocast(x_tokens)) and x_tokens.dim() == 4 and isinstance(conv, torch.nn.Conv2d) \
        and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 \
        and conv.in_channels % 4 == 0 and conv.out_channels % 4 == 0 and conv.weight.dtype == torch.float32 \
        and (conv.bias is None or conv.bias.dtype == torch.float32) and x_tokens.numel() // x_tokens.shape[-1] >= 64


def pointwise_tokens(conv, x_tokens: torch.Tensor) -> torch.Tensor:
    return _Pointwise1x1Fn.apply(_h16(x_tokens).contiguous(), conv.weight, conv.bias)


def patch_embed_ok(conv, x_tokens: torch.Tensor) -> bool:
    """a kernel = stride convolution applied as ONE token Linear to space-to-depth tokens [B, H / p, W / p, C p p] (fp32 device step)"""
    return _fp32_device(x_tokens) and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == conv.stride and conv.groups == 1 \
        and conv.padding == (0, 0) and conv.weight.dtype == torch.float32 and conv.weight[0].numel() % 4 == 0 \
        and conv.out_channels % 4 == 0 and x_tokens.shape[-1] == conv.weight[0].numel() \
        and x_tokens.numel() // x_tokens.shape[-1] >= 64


# ---- 1x1 convolutions to a handful of channels: the side heads and the fuse convolution (csrc/sepconv32.hip head1x1_*) ---------------
def _layout(x: torch.Tensor):
    """(storage tensor, xsb, xsp, xsk, token_major) of an NCHW-LOGICAL tensor x [B, K, H, W] without copying when it is a permuted view
    of token-major storage (what the stages hand over) or plain NCHW"""
    B, K, H, W = x.shape
    P = H * W
    xt = x.permute(0, 2, 3, 1)
    if xt.is_contiguous():
        return xt, P * K, K, 1, True
    xc = x.contiguous()
    return xc, K * P, 1, P, False


class _Head1x1Fn(torch.autograd.Function):
    """y = conv1x1(x) for a conv with <= 8 output channels; x NCHW-logical (either memory layout), y NCHW contiguous.  fp32 (SwT2Net's
    device step) or fp16 activations with the fp32 master weights (the fuse convolution of the fp16-autocast nets: autocast's output
    type, fp32 sums)"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, K, H, W = x.shape
        N, P = weight.shape[0], H * W
        xs, xsb, xsp, xsk, tm = _layout(x)
        sfx = "f16" if xs.dtype == torch.float16 else "f32"
        y = torch.empty((B, N, H, W), dtype=xs.dtype, device=x.device)
        call("nnz_head1x1_forward_" + sfx, ptr(xs), ptr(weight), ptr(bias), ptr(y), B, N, K, P, xsb, xsp, xsk, stream_ptr())
        ctx.save_for_backward(xs, weight)
        ctx.meta = (B, N, K, H, W, xsb, xsp, xsk, tm, bias is not None, sfx)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, weight = ctx.saved_tensors
        B, N, K, H, W, xsb, xsp, xsk, tm, has_bias, sfx = ctx.meta
        P = H * W
        dy = dy.contiguous() if dy.dtype == xs.dtype else dy.to(xs.dtype).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dxs = torch.empty_like(xs)
            call("nnz_head1x1_dgrad_" + sfx, ptr(dy), ptr(weight), ptr(dxs), B, N, K, P, xsb, xsp, xsk, stream_ptr())
            dx = dxs.permute(0, 3, 1, 2) if tm else dxs
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            ws = torch.empty(int(_lib.load().nnz_head1x1_wgrad_workspace_floats(B, N, K, P)), dtype=torch.float32, device=dy.device)
            dwb = torch.empty((N, K + 1), dtype=torch.float32, device=dy.device)
            call("nnz_head1x1_wgrad_" + sfx, ptr(xs), ptr(dy), ptr(ws), ptr(dwb), B, N, K, P, xsb, xsp, xsk, stream_ptr())
            dw = dwb[:, :K].reshape(weight.shape)
            db = dwb[:, K].contiguous() if has_bias else None
        return dx, dw, db


USE_HEAD1X1 = os.environ.get("NNZ_HEAD1X1", "1") != "0"        # A/B switch


def head1x1_ok(conv, x: torch.Tensor) -> bool:
    return USE_HEAD1X1 and (_fp32_device(x) or _h16_autocast(x)) and x.dim() == 4 and isinstance(conv, torch.nn.Conv2d) \
        and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 \
        and conv.out_channels <= 8 and conv.out_channels * conv.in_channels <= 8192 and conv.weight.dtype == torch.float32 \
        and (conv.bias is None or conv.bias.dtype == torch.float32)


def head1x1(conv, x: torch.Tensor) -> torch.Tensor:
    return _Head1x1Fn.apply(_h16(x), conv.weight, conv.bias)
