"""Building blocks shared by the X^2-Net zoo models (M2Net / SS2D^2Net and SwT2Net): the U^2-Net residual conv
blocks, patch merge/expand and small helpers.  Same class names, constructor arguments, sub-module names and
parameter shapes as the reference so that state_dicts interchange:
  REBNCONV, RSU4F            /root/reference/nnunetv2/nets/m2net.py:18-30, 769-801 (== swt2net.py:17-31, 873-905)
  PatchMerging2D, PatchExpand  m2net.py:228-319
  _upsample_like             m2net.py:33-36 (bilinear, align_corners=False)
  Convolution(conv_only)     monai.networks.blocks.Convolution as used at swt2net.py:1058-1066: an nn.Sequential whose
                             only child is named `conv`
  DropPath                   timm.layers.DropPath (m2net) - per-sample stochastic depth
These are thin wrappers over library ops (cuDNN/MIOpen-class convs, rocBLAS GEMMs, LayerNorm); the hand-written
kernels of the zoo path are the selective scan and the window-attention core.
"""
from __future__ import annotations

import os

import torch

from .. import backends as _backends
import torch.nn.functional as F
from torch import nn

from ..token_linear import TokenLinear
from ..layer_norm import LayerNorm


class DropPath(nn.Module):
    def __init__(self, drop_prob: float = 0., scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


class _ResidualDropPathFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, x, mask, scale, rand_keep=0.0):
        """rand_keep > 0: `mask` holds fp32 uniform draws and the 0 / 1 mask is floor(mask + rand_keep), made in the kernel"""
        from .._lib import call, ptr, stream_ptr
        inp, x = inp.contiguous(), x.contiguous()
        B = x.shape[0]
        P = x.numel() // B
        out_dtype = torch.float16 if (inp.dtype == torch.float16 and x.dtype == torch.float16) else torch.float32
        out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        h = torch.float16
        if rand_keep > 0.0:
            assert mask.dtype == torch.float32 and mask.is_contiguous() and mask.numel() == B
            call("nnz_residual_droppath_rand_forward", ptr(inp), int(inp.dtype == h), ptr(x), int(x.dtype == h), ptr(mask),
                 float(rand_keep), float(scale), ptr(out), int(out_dtype == h), B, P, stream_ptr())
        else:
            call("nnz_residual_droppath_forward", ptr(inp), int(inp.dtype == h), ptr(x), int(x.dtype == h), ptr(mask),
                 int(mask is not None and mask.dtype == h), float(scale), ptr(out), int(out_dtype == h), B, P,
                 stream_ptr())
        ctx.save_for_backward(mask)
        ctx.meta = (scale, x.dtype, inp.dtype, B, P)
        ctx.rand_keep = float(rand_keep)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .._lib import call, ptr, stream_ptr
        (mask,) = ctx.saved_tensors
        scale, xdt, idt, B, P = ctx.meta
        dout = dout.contiguous()
        h = torch.float16
        dx = None
        if ctx.needs_input_grad[1] and mask is None and scale == 1.0 and dout.dtype == xdt:
            dx = dout                                   # plain residual: both branches receive the same gradient tensor
        elif ctx.needs_input_grad[1]:
            dx = torch.empty(dout.shape, dtype=xdt, device=dout.device)
            if ctx.rand_keep > 0.0:
                call("nnz_residual_droppath_rand_backward", ptr(dout), int(dout.dtype == h), ptr(mask), ctx.rand_keep,
                     float(scale), ptr(dx), int(xdt == h), B, P, stream_ptr())
            else:
                call("nnz_residual_droppath_backward", ptr(dout), int(dout.dtype == h), ptr(mask),
                     int(mask is not None and mask.dtype == h), float(scale), ptr(dx), int(xdt == h), B, P, stream_ptr())
        dinp = (dout if dout.dtype == idt else dout.to(idt)) if ctx.needs_input_grad[0] else None
        return dinp, dx, None, None, None


def residual_drop_path(inp: torch.Tensor, x: torch.Tensor, drop_path: "DropPath") -> torch.Tensor:
    """`inp + drop_path(x)` (the residual of the VSS / SSND blocks) in one pass each way.  The per-sample mask is drawn
    with the same call as timm's DropPath (same RNG stream); mask scaling, multiply and add are one kernel."""
    ok = x.is_cuda and inp.shape == x.shape and x.dtype in (torch.float16, torch.float32) \
        and inp.dtype in (torch.float16, torch.float32) and x.shape[0] <= 65535 and (x.numel() // x.shape[0]) % 4 == 0
    if not ok:
        return inp + drop_path(x)
    if drop_path.drop_prob == 0. or not drop_path.training:
        return _ResidualDropPathFn.apply(inp, x, None, 1.0)
    keep = 1 - drop_path.drop_prob
    scale = 1.0 / keep if (keep > 0.0 and drop_path.scale_by_keep) else 1.0
    from ..droppath_draws import _ACTIVE, uniform
    if _ACTIVE and keep > 0.0:
        # a row of the pass's draw table (ONE torch.rand per forward, droppath_draws.py); the 0 / 1 mask floor(keep + u) - a
        # Bernoulli(keep) variable like timm's bernoulli_(keep) below - is formed inside the residual kernel
        return _ResidualDropPathFn.apply(inp, x, uniform(x.shape[0], x.device), scale, keep)
    mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
    return _ResidualDropPathFn.apply(inp, x, mask, scale)


class _DepthwiseNativeFn(torch.autograd.Function):
    """fp32 depthwise convolution through ATen's native depthwise kernels instead of the library path.  Measured on the
    SwT2Net step (tools/probes/swt_slow_conv_probe.py): MIOpen runs the weight gradient of the three fp32 depthwise 3x3
    convolutions (32 ch @ 512^2, 64 @ 256^2, 128 @ 128^2) as a batched xdlops GEMM - 72 + 19 + 5 ms of a 206 ms step;
    ATen's direct kernels take < 3 ms for the same three.  Under fp16 autocast (SSND2Net) the library picks a 24 ms grouped-conv
    weight-gradient kernel for the same layers: 379 -> 277 ms per step with the direct kernels.  The backend is picked again inside convolution_backward, so the
    switch has to wrap the backward call too - hence an autograd Function rather than a context manager in forward."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dilation, groups):
        # a permuted token tensor (the U^2 stages hand channels-last views to the depthwise stem of the next stage) is packed HERE:
        # ATen's kernel would pack it anyway, and the saved copy keeps the weight gradient on csrc/depthwise_wgrad.hip (round 5:
        # the four decoder-side stems of SwT2Net fell to ATen's 580 us weight-gradient kernel for want of a contiguous input)
        x = x.contiguous()
        with torch.backends.cudnn.flags(enabled=False):
            y = F.conv2d(x, w, b, stride, padding, dilation, groups)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, dilation, groups, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, dilation, groups, has_b = ctx.cfg
        dy = dy.contiguous()
        need_w, need_b = ctx.needs_input_grad[1], has_b and ctx.needs_input_grad[2]
        dw = db = None
        if (need_w or need_b) and _dw_wgrad_ok(x, dy, w, stride, padding, dilation):
            # weight / bias gradient on csrc/depthwise_wgrad.hip (ATen's kernel gives each (channel, tap) one workgroup
            # over the whole batch: 110-126 us per call in the SSND2Net / LightMamba2Net steps)
            from .._lib import call, load, ptr, stream_ptr
            B, C, H, W = x.shape
            buf = torch.empty(load().nnz_dwconv2d_wgrad_workspace_floats(B, C, H, W) + C * 10, dtype=torch.float32,
                              device=x.device)
            dw32, db32, ws = buf[:C * 9], buf[C * 9:C * 10], buf[C * 10:]
            call("nnz_dwconv2d_wgrad", ptr(x), ptr(dy), int(x.dtype == torch.float16), ptr(ws), ptr(dw32),
                 ptr(db32) if need_b else None, B, C, H, W, int(dilation[0]), stream_ptr())
            if w.dtype != torch.float32:   # fp16 parameter shadow (param_shadow.py): ONE cast launch for weight + bias gradient
                lo = buf[:C * 10].to(w.dtype)
                dw32, db32 = lo[:C * 9], lo[C * 9:]
            if w.shape[-1] == 1:       # 1x1 depthwise (a per-channel scale): the centre tap of the 3x3 sums
                dw32 = dw32.view(C, 9)[:, 4]
            dw = dw32.reshape(w.shape) if need_w else None
            db = db32 if need_b else None
            need_w = need_b = False
        dx = dw2 = db2 = None
        if ctx.needs_input_grad[0] or need_w or need_b:
            with torch.backends.cudnn.flags(enabled=False):
                dx, dw2, db2 = torch.ops.aten.convolution_backward(
                    dy, x, w, [w.shape[0]] if has_b else None, list(stride), list(padding), list(dilation), False,
                    [0, 0], groups, [ctx.needs_input_grad[0], need_w, need_b])
        return dx, (dw if dw is not None else dw2), (db if db is not None else db2), None, None, None, None


def _dw_wgrad_ok(x, dy, w, stride, padding, dilation, will_pack: bool = False) -> bool:
    """will_pack: asked at dispatch time, before _DepthwiseNativeFn.forward has made the input contiguous"""
    k3 = tuple(w.shape[1:]) == (1, 3, 3) and dilation[0] == dilation[1] and tuple(padding) == tuple(dilation)
    k1 = tuple(w.shape[1:]) == (1, 1, 1) and tuple(padding) == (0, 0) and tuple(dilation) == (1, 1)
    return (os.environ.get("NNZ_DW_WGRAD", "1") != "0" and x.dim() == 4 and (k3 or k1) and tuple(stride) == (1, 1)
            and x.dtype == dy.dtype and x.dtype in (torch.float16, torch.float32) and (will_pack or x.is_contiguous())
            and dy.shape == x.shape and x.shape[1] <= 65535)


class _Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters / state_dict) that sends depthwise calls on the GPU to _DepthwiseNativeFn"""

    def forward(self, x):
        mode = os.environ.get("NNZ_DW_NATIVE", "2")     # 0: library path, 1: fp32 calls only, 2: fp16 autocast calls too
        if (self.groups == self.in_channels == self.out_channels and self.groups > 1 and x.is_cuda
                and self.padding_mode == "zeros" and mode != "0"):
            w, b = self.weight, self.bias
            if torch.is_autocast_enabled():
                if mode != "2":
                    _backends.note(self, "library", why="NNZ_DW_NATIVE=1 under autocast")
                    return super().forward(x)
                dt = torch.get_autocast_dtype("cuda")   # what autocast would have cast the convolution's operands to
                x, w, b = x.to(dt), w.to(dt), (b.to(dt) if b is not None else None)
            elif x.dtype != torch.float32:
                _backends.note(self, "library", why=f"depthwise input {x.dtype} outside autocast")
                return super().forward(x)
            # forward / input gradient: ATen's direct depthwise kernels (chosen over MIOpen's batched-GEMM path); weight
            # gradient: csrc/depthwise_wgrad.hip where _dw_wgrad_ok
            _backends.note(self, "aten")
            _backends.note(self, "hip" if _dw_wgrad_ok(x, x, w, self.stride, self.padding, self.dilation, will_pack=True)
                           else "aten", site="wgrad")
            return _DepthwiseNativeFn.apply(x, w, b, self.stride, self.padding, self.dilation, self.groups)
        if self.groups == self.in_channels == self.out_channels and self.groups > 1 and x.is_cuda:
            _backends.note(self, "library", why="depthwise with NNZ_DW_NATIVE=0 or non-zero padding mode")
        return super().forward(x)


class Convolution(nn.Sequential):
    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, strides=1, kernel_size=3, bias=True,
                 conv_only: bool = True, groups: int = 1, dilation: int = 1, padding=None):
        super().__init__()
        if not conv_only:
            raise NotImplementedError("only the conv_only form is used by the zoo")
        conv = {2: _Conv2d, 3: nn.Conv3d}[spatial_dims]
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        pad = (k - 1) // 2 * dilation if padding is None else padding
        self.add_module("conv", conv(in_channels, out_channels, kernel_size, strides, pad, dilation, groups, bias))


def get_dwconv_layer(spatial_dims: int, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1,
                     bias: bool = False):
    depth = Convolution(spatial_dims, in_channels, in_channels, strides=stride, kernel_size=kernel_size, bias=bias,
                        conv_only=True, groups=in_channels)
    point = Convolution(spatial_dims, in_channels, out_channels, strides=stride, kernel_size=1, bias=bias,
                        conv_only=True, groups=1)
    return nn.Sequential(depth, point)


_INTERP_MATRICES = {}


def _interp_matrix(n_in: int, n_out: int, device) -> torch.Tensor:
    """(n_out, n_in) matrix of 1-D linear interpolation with align_corners=False - torch's source index rule
    max((o + 0.5) * n_in / n_out - 0.5, 0), neighbours clamped to the last sample"""
    key = (n_in, n_out, str(device))
    if key not in _INTERP_MATRICES:
        o = torch.arange(n_out, dtype=torch.float32, device=device)
        src = ((o + 0.5) * (float(n_in) / float(n_out)) - 0.5).clamp_min(0)
        i0 = src.floor().long().clamp_max(n_in - 1)
        i1 = (i0 + 1).clamp_max(n_in - 1)
        w1 = src - i0.float()
        W = torch.zeros(n_out, n_in, dtype=torch.float32, device=device)
        W.scatter_add_(1, i0[:, None], (1 - w1)[:, None])
        W.scatter_add_(1, i1[:, None], w1[:, None])
        _INTERP_MATRICES[key] = W
    return _INTERP_MATRICES[key]


USE_HIP_UPSAMPLE = os.environ.get("NNZ_UPSAMPLE_HIP", "1") != "0"      # A/B switch: 0 = torch forward + two-GEMM adjoint (rounds 2-5)


class _BilinearUpFn(torch.autograd.Function):
    """F.interpolate(bilinear, align_corners=False) and its backward as the ADJOINT of the interpolation written as a gather
    (csrc/upsample.hip: every input pixel sums the output pixels that read it, fixed order).  ATen's backward scatters every output
    pixel into its 4 sources with atomics: for the 32x side outputs of the X^2-Nets that is 438 us per call on a 4 MB tensor
    (profiles/r02_m2net_graph_kernels.txt: 1.7 ms of the 101 ms M2Net step, the same in SwT2Net).  Rounds 2-5: dIn = Wy^T dOut Wx as two
    small library GEMMs behind torch's own forward - still available as NNZ_UPSAMPLE_HIP=0."""

    @staticmethod
    def forward(ctx, src, size):
        hip = USE_HIP_UPSAMPLE and src.is_cuda and src.dtype in (torch.float16, torch.float32) and src.dim() == 4 \
            and src.shape[0] * src.shape[1] <= 65535 and size[1] <= 16384
        ctx.meta = (tuple(src.shape[2:]), tuple(size), src.dtype, hip)
        if not hip:
            return F.interpolate(src, size=size, mode='bilinear', align_corners=False)
        from .._lib import call, ptr, stream_ptr
        sc = src.contiguous()
        N, Cc, h, w = sc.shape
        out = torch.empty((N, Cc, size[0], size[1]), dtype=sc.dtype, device=sc.device)
        call("nnz_bilinear_up_forward", ptr(sc), ptr(out), int(sc.dtype == torch.float16), N * Cc, h, w, size[0], size[1], stream_ptr())
        return out

    @staticmethod
    def backward(ctx, g):
        (h, w), (H, W), dtype, hip = ctx.meta
        if hip:
            from .._lib import call, ptr, stream_ptr
            gc = (g if g.dtype == dtype else g.to(dtype)).contiguous()
            N, Cc = gc.shape[:2]
            gin = torch.empty((N, Cc, h, w), dtype=dtype, device=g.device)
            call("nnz_bilinear_up_backward", ptr(gc), ptr(gin), int(dtype == torch.float16), N * Cc, h, w, H, W, stream_ptr())
            return gin, None
        with torch.autocast("cuda", enabled=False):
            Wy, Wx = _interp_matrix(h, H, g.device), _interp_matrix(w, W, g.device)
            gin = torch.matmul(torch.matmul(Wy.t(), g.float()), Wx)
        return gin.to(dtype), None


def _upsample_like(src, tar_shape):
    tar_shape = tuple(int(v) for v in tar_shape)
    if src.is_cuda and src.dim() == 4 and src.requires_grad and os.environ.get("NNZ_UPSAMPLE_ADJOINT", "1") != "0":
        return _BilinearUpFn.apply(src, tar_shape)
    return F.interpolate(src, size=tar_shape, mode='bilinear', align_corners=False)


class REBNCONV(nn.Module):
    def __init__(self, in_ch=3, out_ch=3, dirate=1):
        super().__init__()
        self.conv_s1 = nn.Conv2d(in_ch, out_ch, 3, padding=dirate, dilation=dirate)
        self.bn_s1 = nn.BatchNorm2d(out_ch)
        self.relu_s1 = nn.ReLU(inplace=True)

    def forward(self, x):
        from .. import rebnconv
        if rebnconv.unit_ok(self, x):
            # on its own (the `rebnconvin` of an MU stage): the same tap-table conv + batch-stat norm + ReLU kernels as inside an RSU4F;
            # the result is an NCHW view of channels-last fp16 storage
            _backends.note(self, "hip")
            return rebnconv.unit_nchw(self, x)
        return self.relu_s1(self.bn_s1(self.conv_s1(x)))


class RSU4F(nn.Module):
    """dilated residual U block (dilations 1, 2, 4, 8 and back).  `block` is the conv-BN-ReLU unit: the dilated
    REBNCONV of m2net.py, or the depthwise-separable (dilation-free) one swt2net.py:17-31 defines under the same name."""
    block = REBNCONV

    def __init__(self, in_ch=3, mid_ch=12, out_ch=3):
        super().__init__()
        B = self.block
        self.rebnconvin = B(in_ch, out_ch, dirate=1)
        self.rebnconv1 = B(out_ch, mid_ch, dirate=1)
        self.rebnconv2 = B(mid_ch, mid_ch, dirate=2)
        self.rebnconv3 = B(mid_ch, mid_ch, dirate=4)
        self.rebnconv4 = B(mid_ch, mid_ch, dirate=8)
        self.rebnconv3d = B(mid_ch * 2, mid_ch, dirate=4)
        self.rebnconv2d = B(mid_ch * 2, mid_ch, dirate=2)
        self.rebnconv1d = B(mid_ch * 2, out_ch, dirate=1)

    def forward(self, x):
        from .. import rebnconv, sepconv32
        if sepconv32.hip_path_ok(self, x):
            # the depthwise-separable fp32 unit of swt2net.py: depthwise 3x3, pointwise 1x1 on the fp32 MFMA Linear kernels, batch-stat
            # norm + ReLU - token-major inside the block, no library call (nnuzoo_amd/sepconv32.py)
            _backends.note(self, "hip-f32")
            return sepconv32.rsu4f_forward(self, x)
        _backends.note(self, "hip" if rebnconv.USE_HIP and rebnconv.hip_path_ok(self, x) else "library",
                       why="RSU4F outside fp16 autocast / unsupported channels / eval with autograd")
        if rebnconv.USE_HIP and rebnconv.hip_path_ok(self, x):
            # dilated conv + batch-stat norm + ReLU units on the tap-table conv kernels, channels-last inside the block
            return rebnconv.rsu4f_forward(self, x)
        xin = self.rebnconvin(x)
        e1 = self.rebnconv1(xin)
        e2 = self.rebnconv2(e1)
        e3 = self.rebnconv3(e2)
        e4 = self.rebnconv4(e3)
        d3 = self.rebnconv3d(torch.cat((e4, e3), 1))
        d2 = self.rebnconv2d(torch.cat((d3, e2), 1))
        d1 = self.rebnconv1d(torch.cat((d2, e1), 1))
        return d1 + xin


class PatchMerging2D(nn.Module):
    """scale x scale space-to-depth -> LayerNorm -> Linear (token-major in/out; `permute` for NCHW callers)."""

    def __init__(self, input_dim: int, scale: int, output_features: int = None, norm_layer=LayerNorm):
        super().__init__()
        self.input_feature_size = (scale ** 2) * input_dim
        self.output_features = output_features or input_dim * scale
        self.scale = scale
        self.reduction = TokenLinear(self.input_feature_size, self.output_features, bias=False)
        self.norm = norm_layer(self.input_feature_size)
        if isinstance(self.norm, LayerNorm):
            self.norm.feeds_linear = True     # its only consumer is `reduction`: fp16 rows straight from the kernel under autocast

    def forward(self, x, permute=False):
        if permute:
            x = x.permute(0, 2, 3, 1)
        B, H, W, C = x.shape
        s = self.scale
        Hs, Ws = H // s, W // s
        if s != 2:
            raise NotImplementedError("the zoo only merges 2x2 patches")
        # channel order of the reference: (0,0), (1,0), (0,1), (1,1)  (m2net.py:254-267)
        # = torch.cat of the four strided slices, as ONE permuted copy: channel block k = 2 * (column parity) + (row parity);
        # the backward is one strided copy instead of four zero fills, four slice copies and three adds
        x = x[:, :2 * Hs, :2 * Ws].unflatten(2, (Ws, 2)).unflatten(1, (Hs, 2)).permute(0, 1, 3, 4, 2, 5) \
            .reshape(B, Hs, Ws, 4 * C)
        x = self.reduction(self.norm(x))
        if permute:
            x = x.permute(0, 3, 1, 2).contiguous()
        return x


class PatchExpand(nn.Module):
    """NCHW in -> token-major out.  output_dim None: Linear(dim -> scale*dim) then depth-to-space (dim/scale ch);
    output_dim given: depth-to-space first (dim/scale^2 ch) then Linear to output_dim.  LayerNorm last."""

    def __init__(self, dim: int, scale, output_dim: int = None, norm_layer=LayerNorm):
        super().__init__()
        self.dim, self.scale, self.output_dim = dim, scale, output_dim
        if output_dim is None:
            self.expand = TokenLinear(dim, scale * dim, bias=False)
            self.norm = norm_layer(dim // scale)
        else:
            self.expand = TokenLinear(dim // (scale ** 2), output_dim, bias=False)
            self.norm = norm_layer(output_dim)

    def _d2s(self, x):
        B, H, W, C = x.shape
        s = self.scale
        c = C // (s * s)
        return x.view(B, H, W, s, s, c).permute(0, 1, 3, 2, 4, 5).reshape(B, H * s, W * s, c)

    def forward(self, x, permute=False):
        x = x.permute(0, 2, 3, 1)
        if self.output_dim is None:
            x = self._d2s(self.expand(x))
        else:
            x = self.expand(self._d2s(x))
        x = self.norm(x)
        if permute:
            x = x.permute(0, 3, 1, 2).contiguous()
        return x
