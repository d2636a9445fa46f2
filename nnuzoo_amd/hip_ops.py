"""Thin torch-tensor wrappers over the C-ABI (include/nnuzoo_hip.h).  No arithmetic happens here: every
function validates dtype/device, hands raw device pointers + the current HIP stream to libnnuzoo_hip.so and
raises if the library reports an error.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import call, ptr, stream_ptr
from .conv_plan import TapTable, ksel_array


def _f16(t: torch.Tensor, name: str):
    if t.dtype != torch.float16 or not t.is_cuda:
        raise _lib.HipCallError(f"{name}: expected a float16 device tensor, got {t.dtype} on {t.device}")


def _f32(t: Optional[torch.Tensor], name: str):
    if t is not None and (t.dtype != torch.float32 or not t.is_cuda):
        raise _lib.HipCallError(f"{name}: expected a float32 device tensor, got {t.dtype} on {t.device}")


class PreparedTable:
    """TapTable + its ctypes descriptor (built once, reused every step)."""

    def __init__(self, table: TapTable):
        self.table = table
        self.desc = table.to_desc()
        self.pack_ksel = ksel_array(table.pack_ksel)
        self.ident_ksel = ksel_array(list(range(32)))
        m = table.m_dims[0] * table.m_dims[1] * table.m_dims[2]
        # algorithmic FLOPs of one launch (2 * MAC): every m-voxel x tap x Cin x Cout
        self.flops = 2.0 * table.N * m * table.ntaps * table.Cin * table.Cout

    def with_accumulate(self, acc: bool) -> "PreparedTable":
        import copy
        t = copy.copy(self.table)
        t.accumulate = acc
        return PreparedTable(t)


class LaunchTimer:
    """HIP-event timing of the dominant kernels' launches on the stream they run on (bench.py's roofline legs).  `work`
    is the algorithmic work of the launch: FLOPs for the MFMA-bound conv kernels, HBM bytes for the scan kernels."""

    def __init__(self):
        self.enabled = False
        self.records = []  # (kind, work, start, end)

    def wrap(self, kind: str, flops: float, fn):
        if not self.enabled:
            fn()
            return
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.records.append((kind, flops, s, e))

    def summary(self):
        """{kind: (launches, total_flops, total_seconds)} - call after torch.cuda.synchronize()."""
        out = {}
        for kind, fl, s, e in self.records:
            n, f, t = out.get(kind, (0, 0.0, 0.0))
            out[kind] = (n + 1, f + fl, t + s.elapsed_time(e) * 1e-3)
        return out


TIMER = LaunchTimer()


def pack_weight(param: torch.Tensor, pt: PreparedTable, R: int, Cc: int, sr: int, sc: int, sk: int,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 torch parameter -> packed fp16 [R/16][Cc/32][T][32][16] (R = reduction channels, Cc = output channels)."""
    _f32(param, "pack_weight.param")
    T = pt.table.ntaps
    if out is None:
        out = torch.empty(R * Cc * T, dtype=torch.float16, device=param.device)
    call("nnz_pack_conv_weight", ptr(param), ptr(out), R, Cc, T, sr, sc, sk, pt.pack_ksel, stream_ptr())
    return out


def conv_tap_forward(pt: PreparedTable, x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor],
                     out: torch.Tensor, stats: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None,
                     innorm: Optional["InNorm"] = None) -> None:
    """stats: optional pre-zeroed fp32 [N, Cout, 2]; the kernel adds {sum, sumsq} of its fp16 outputs (fused
    InstanceNorm statistics).  workspace: fp32 scratch that lets few-tile / long-reduction layers run split-K.
    innorm: x is a raw conv output, normalised + activated while it is staged"""
    _f16(x, "conv.in"); _f16(out, "conv.out"); _f16(w_packed, "conv.w"); _f32(bias, "conv.bias")
    _f32(stats, "conv.stats"); _f32(workspace, "conv.workspace")
    if innorm is not None:
        assert stats is None
        TIMER.wrap("conv_box_kernel", pt.flops,
                   lambda: call("nnz_conv_tap_forward_innorm", ptr(x), ptr(out), ptr(w_packed), ptr(bias), C.byref(pt.desc),
                                ptr(innorm.tab), innorm.c0, innorm.slope, None, None, None, None, 0.0, None,
                                ptr(workspace), 0 if workspace is None else workspace.numel(), stream_ptr()))
        return
    if workspace is not None and stats is None:
        TIMER.wrap("conv_box_kernel", pt.flops,
                   lambda: call("nnz_conv_tap_forward_ws", ptr(x), ptr(out), ptr(w_packed), ptr(bias), C.byref(pt.desc),
                                ptr(workspace), workspace.numel(), stream_ptr()))
        return
    TIMER.wrap("conv_box_kernel", pt.flops,
               lambda: call("nnz_conv_tap_forward_stats", ptr(x), ptr(out), ptr(w_packed), ptr(bias),
                            C.byref(pt.desc), ptr(stats), stream_ptr()))


class InNorm:
    """Consumer-side InstanceNorm + LeakyReLU of an operand (include/nnuzoo_hip.h, round 4): the operand is the RAW conv output
    of its producer block; `tab` is that block's table [N, C - c0, 4] = {mean, rstd, scale, shift}; channels [0, c0) pass
    unchanged (the transposed-conv half of a cat buffer)."""
    __slots__ = ("tab", "c0", "slope")

    def __init__(self, tab: torch.Tensor, slope: float, c0: int = 0):
        self.tab, self.c0, self.slope = tab, int(c0), float(slope)     # (validated by the launch wrappers: ptr() needs HIP memory)


class NormScratch:
    """Per-device scratch of the deterministic statistics (csrc/common.hpp FxAcc): fixed-point accumulators for the
    widest (N, C) in use and the launch counter.  Zero-initialised ONCE; every launch leaves them zero again, so the same
    scratch serves all conv blocks of a network one after the other (stream order)."""

    def __init__(self, device, n_times_c: int):
        nb = int(_lib.load().nnz_fxacc_bytes())
        n_times_c = max(int(n_times_c), 2816)   # also the stem (864) and seg-head (8 * 641) weight-gradient sums
        # (room for 4 records per unit of capacity: the conv epilogue keeps 2 or - with the matrix-core moments - 3 per (n, c))
        self.acc = torch.zeros(n_times_c * 4 * nb // 8, dtype=torch.int64, device=device)
        self.counter = torch.zeros(2, dtype=torch.int32, device=device)
        self.capacity = n_times_c
        self.records = 2 * n_times_c


_SCRATCH = {}
_SCRATCH_RETIRED = []   # outgrown scratches stay alive: captured hipGraphs have their `acc` / `counter` pointers baked in


def det_scratch(device, records: int = 0) -> NormScratch:
    """The device's shared scratch of the deterministic reductions (loss sums, gradient norm, ...).  Single-stream contract:
    launches on ONE stream use it one after the other and each leaves it zeroed (the last-workgroup ticket and the records are
    shared, so two streams using it concurrently would corrupt each other's sums).  Grows on demand; an outgrown scratch is
    never freed (a captured graph may still write to it) and growing while a stream is capturing is refused - the capture
    would record pointers of a buffer whose zero-initialisation is not part of the graph."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    sc = _SCRATCH.get(key)
    if sc is None or sc.capacity * 2 < records:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"det_scratch: {records} records requested during hipGraph capture but the device scratch "
                               f"holds {0 if sc is None else sc.capacity * 2}; run one eager warm-up pass first")
        if sc is not None:
            _SCRATCH_RETIRED.append(sc)
        sc = NormScratch(device, max((records + 1) // 2, 4096))
        _SCRATCH[key] = sc
    return sc


def conv_tap_forward_norm(pt: PreparedTable, x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor],
                          out: torch.Tensor, scratch: NormScratch, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                          nstat: torch.Tensor, workspace: Optional[torch.Tensor] = None,
                          innorm: Optional["InNorm"] = None) -> None:
    """forward convolution + the InstanceNorm table of its output: nstat [N, Cout, 4] = {mean, rstd, scale, shift}
    (deterministic fixed-point statistics, written by the launch's last workgroup).  innorm: x is the RAW output of the
    producer block(s), normalised + activated while the input box is staged (consumer side of conv + norm + act)"""
    _f16(x, "conv.in"); _f16(out, "conv.out"); _f16(w_packed, "conv.w"); _f32(bias, "conv.bias")
    _f32(gamma, "conv.gamma"); _f32(beta, "conv.beta"); _f32(nstat, "conv.nstat")
    assert pt.table.N * pt.table.Cout <= scratch.capacity
    _f32(workspace, "conv.workspace")
    if innorm is not None:
        TIMER.wrap("conv_box_kernel", pt.flops,
                   lambda: call("nnz_conv_tap_forward_innorm", ptr(x), ptr(out), ptr(w_packed), ptr(bias), C.byref(pt.desc),
                                ptr(innorm.tab), innorm.c0, innorm.slope, ptr(scratch.acc), ptr(scratch.counter), ptr(gamma),
                                ptr(beta), float(eps), ptr(nstat), ptr(workspace),
                                0 if workspace is None else workspace.numel(), stream_ptr()))
        return
    TIMER.wrap("conv_box_kernel", pt.flops,
               lambda: call("nnz_conv_tap_forward_norm_ws", ptr(x), ptr(out), ptr(w_packed), ptr(bias), C.byref(pt.desc),
                            ptr(scratch.acc), ptr(scratch.counter), ptr(gamma), ptr(beta), float(eps), ptr(nstat),
                            ptr(workspace), 0 if workspace is None else workspace.numel(), stream_ptr()))


def convT_supported(cin: int, cout: int, stride, dgrad: bool) -> bool:
    return bool(_lib.load().nnz_convT_supported(cin, cout, int(stride[0]), int(stride[1]), int(stride[2]), int(dgrad)))


def convT_forward(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, N: int, in_dims,
                  cin: int, cout: int, stride, ldi: int, ldo: int, innorm: Optional["InNorm"] = None) -> None:
    """kernel = stride ConvTranspose on its own HBM-bound kernel (csrc/conv_transpose.hip); weight = the fp32 parameter"""
    _f16(x, "convT.in"); _f16(out, "convT.out"); _f32(weight, "convT.w"); _f32(bias, "convT.bias")
    flops = 2.0 * N * in_dims[0] * in_dims[1] * in_dims[2] * cin * cout * stride[0] * stride[1] * stride[2]
    if innorm is not None:
        assert innorm.c0 == 0
        TIMER.wrap("convT_kernel", flops, lambda: call(
            "nnz_convT_forward_innorm", ptr(x), ptr(innorm.tab), innorm.slope, ptr(weight), ptr(bias), ptr(out), N,
            int(in_dims[0]), int(in_dims[1]), int(in_dims[2]), cin, cout, int(stride[0]), int(stride[1]), int(stride[2]), ldi,
            ldo, stream_ptr()))
        return
    TIMER.wrap("convT_kernel", flops, lambda: call(
        "nnz_convT_forward", ptr(x), ptr(weight), ptr(bias), ptr(out), N, int(in_dims[0]), int(in_dims[1]), int(in_dims[2]),
        cin, cout, int(stride[0]), int(stride[1]), int(stride[2]), ldi, ldo, stream_ptr()))


def convT_dgrad(dout: torch.Tensor, weight: torch.Tensor, din: torch.Tensor, N: int, in_dims, cin: int, cout: int, stride,
                ldi: int, ldo: int) -> None:
    _f16(dout, "convT.dout"); _f16(din, "convT.din"); _f32(weight, "convT.w")
    flops = 2.0 * N * in_dims[0] * in_dims[1] * in_dims[2] * cin * cout * stride[0] * stride[1] * stride[2]
    TIMER.wrap("convT_kernel", flops, lambda: call(
        "nnz_convT_dgrad", ptr(dout), ptr(weight), ptr(din), N, int(in_dims[0]), int(in_dims[1]), int(in_dims[2]), cin, cout,
        int(stride[0]), int(stride[1]), int(stride[2]), ldi, ldo, stream_ptr()))


def conv_tap_wgrad(pt: PreparedTable, boxed: torch.Tensor, plain: torch.Tensor, dw: torch.Tensor,
                   pre_zeroed: bool = False) -> None:
    _f16(boxed, "wgrad.boxed"); _f16(plain, "wgrad.plain"); _f32(dw, "wgrad.dw")
    TIMER.wrap("conv_wgrad_kernel", pt.flops,
               lambda: call("nnz_conv_tap_wgrad", ptr(boxed), ptr(plain), ptr(dw), C.byref(pt.desc), int(pre_zeroed),
                            stream_ptr()))


def conv_tap_wgrad_workspace_floats(pt: PreparedTable) -> int:
    return int(_lib.load().nnz_conv_tap_wgrad_workspace_floats(C.byref(pt.desc)))


def conv_tap_wgrad_to_grad(pt: PreparedTable, boxed: torch.Tensor, plain: torch.Tensor, ws: torch.Tensor,
                           grad: torch.Tensor, sa: int, sb: int, sk: int, accumulate: bool = False,
                           boxed_norm: Optional["InNorm"] = None, plain_norm: Optional["InNorm"] = None) -> None:
    """weight gradient straight into the torch-layout `grad` (two-stage, deterministic): partial blocks in `ws`, then a
    fixed-order reduction; grad[a*sa + b*sb + t*sk] with (a, b) = (boxed, plain) channels.  boxed_norm / plain_norm: that
    operand is a raw conv output, normalised + activated while its tile is staged"""
    _f16(boxed, "wgrad.boxed"); _f16(plain, "wgrad.plain"); _f32(ws, "wgrad.ws"); _f32(grad, "wgrad.grad")
    if boxed_norm is not None or plain_norm is not None:
        bn, pn = boxed_norm, plain_norm
        TIMER.wrap("conv_wgrad_kernel", pt.flops,
                   lambda: call("nnz_conv_tap_wgrad_to_grad_innorm", ptr(boxed), ptr(plain), ptr(ws), ws.numel(), ptr(grad), sa,
                                sb, sk, pt.ident_ksel, int(accumulate), C.byref(pt.desc),
                                ptr(bn.tab) if bn else None, bn.c0 if bn else 0, bn.slope if bn else 0.0,
                                ptr(pn.tab) if pn else None, pn.c0 if pn else 0, pn.slope if pn else 0.0, stream_ptr()))
        return
    TIMER.wrap("conv_wgrad_kernel", pt.flops,
               lambda: call("nnz_conv_tap_wgrad_to_grad", ptr(boxed), ptr(plain), ptr(ws), ws.numel(), ptr(grad), sa, sb,
                            sk, pt.ident_ksel, int(accumulate), C.byref(pt.desc), stream_ptr()))


class PackJobTable:
    """Device-resident table of weight-pack jobs (one launch packs all of them)."""

    def __init__(self, device):
        self.device = device
        self.jobs = []
        self.table = None

    def add(self, param: torch.Tensor, dst: torch.Tensor, pt: PreparedTable, R: int, Cc: int, sr: int, sc: int, sk: int):
        self.jobs.append((param, dst, pt, R, Cc, sr, sc, sk))
        self.table = None

    def run(self):
        lib = _lib.load()
        if self.table is None:
            nb = lib.nnz_pack_job_bytes()
            host = (C.c_char * (nb * len(self.jobs)))()
            for i, (param, dst, pt, R, Cc, sr, sc, sk) in enumerate(self.jobs):
                _lib.check(lib.nnz_pack_job_fill(C.byref(host, i * nb), ptr(param), ptr(dst), R, Cc, pt.table.ntaps, sr, sc,
                                                 sk, pt.pack_ksel), "nnz_pack_job_fill")
            self.table = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(self.device)
            self._ptrs = [(j[0].data_ptr(), j[1].data_ptr()) for j in self.jobs]
        else:
            assert self._ptrs == [(j[0].data_ptr(), j[1].data_ptr()) for j in self.jobs], "parameter storage moved"
        call("nnz_pack_conv_weights_batched", ptr(self.table), len(self.jobs), stream_ptr())


class DualPackTable:
    """Device-resident job table of the dual weight pack (csrc/conv_pack.hip pack_dual_kernel): one launch per step reads
    every parameter once and writes the forward AND the data-gradient packed form."""

    def __init__(self, device):
        self.device = device
        self.jobs = []
        self.table = None

    def add(self, param: torch.Tensor, dst_fwd: torch.Tensor, dst_dgrad: torch.Tensor, X: int, Y: int, nk: int,
            x_inner: bool, fwd: PreparedTable, dgrad: PreparedTable):
        assert param.is_contiguous() and param.dtype == torch.float32
        self.jobs.append((param, dst_fwd, dst_dgrad, X, Y, nk, int(x_inner), fwd, dgrad))
        self.table = None

    def run(self):
        if not self.jobs:
            return
        lib = _lib.load()
        if self.table is None:
            nb = lib.nnz_pack_dual_job_bytes()
            host = (C.c_char * (nb * len(self.jobs)))()
            blocks = 0
            for i, (param, df, db, X, Y, nk, xin, fwd, dgrad) in enumerate(self.jobs):
                _lib.check(lib.nnz_pack_dual_job_fill(C.byref(host, i * nb), ptr(param), ptr(df), ptr(db), X, Y, nk, xin,
                                                      blocks, fwd.pack_ksel, dgrad.pack_ksel), "nnz_pack_dual_job_fill")
                blocks += (X // 32) * (Y // 32)
            self.table = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(self.device)
            self.blocks, self.max_nk = blocks, max(j[5] for j in self.jobs)
            self._ptrs = [(j[0].data_ptr(), j[1].data_ptr(), j[2].data_ptr()) for j in self.jobs]
        else:
            assert self._ptrs == [(j[0].data_ptr(), j[1].data_ptr(), j[2].data_ptr()) for j in self.jobs], \
                "parameter storage moved"
        call("nnz_pack_dual_batched", ptr(self.table), len(self.jobs), self.blocks, self.max_nk, stream_ptr())


def unpack_wgrad(dw: torch.Tensor, grad: torch.Tensor, A: int, B: int, T: int, sa: int, sb: int, sk: int,
                 pt: PreparedTable, accumulate: bool = False) -> None:
    _f32(dw, "unpack.dw"); _f32(grad, "unpack.grad")
    call("nnz_unpack_conv_wgrad", ptr(dw), ptr(grad), A, B, T, sa, sb, sk, pt.ident_ksel, int(accumulate),
         stream_ptr())


def stem_forward(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], y: torch.Tensor, dims, ldy: int):
    _f32(x, "stem.x"); _f32(w, "stem.w"); _f32(b, "stem.b"); _f16(y, "stem.y")
    N, D, H, W = dims
    call("nnz_stem_conv_forward", ptr(x), ptr(w), ptr(b), ptr(y), N, D, H, W, w.shape[0], ldy, stream_ptr())


def stem_wgrad(x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, dims, lddy: int, scratch: "NormScratch" = None):
    """scratch: deterministic (fixed-point) cross-workgroup sum instead of float atomics"""
    _f32(x, "stem.x"); _f16(dy, "stem.dy"); _f32(dw, "stem.dw")
    N, D, H, W = dims
    if scratch is None:
        call("nnz_stem_conv_wgrad", ptr(x), ptr(dy), ptr(dw), N, D, H, W, dw.shape[0], lddy, stream_ptr())
    else:
        assert scratch.capacity * 2 >= 864
        call("nnz_stem_conv_wgrad_det", ptr(x), ptr(dy), ptr(dw), N, D, H, W, dw.shape[0], lddy, ptr(scratch.acc),
             ptr(scratch.counter), stream_ptr())


def head_forward(x, w, b, logits, N, V, Cc, K, ldx, innorm: Optional["InNorm"] = None):
    _f16(x, "head.x"); _f32(w, "head.w"); _f32(b, "head.b"); _f16(logits, "head.logits")
    if innorm is not None:
        assert innorm.c0 == 0
        call("nnz_seg_head_forward_innorm", ptr(x), ptr(innorm.tab), innorm.slope, ptr(w), ptr(b), ptr(logits), N, V, Cc, K,
             ldx, stream_ptr())
        return
    call("nnz_seg_head_forward", ptr(x), ptr(w), ptr(b), ptr(logits), N, V, Cc, K, ldx, stream_ptr())


def head_dgrad(dlogits, w, dx, N, V, Cc, K, lddx, accumulate):
    _f16(dlogits, "head.dlogits"); _f32(w, "head.w"); _f16(dx, "head.dx")
    call("nnz_seg_head_dgrad", ptr(dlogits), ptr(w), ptr(dx), N, V, Cc, K, lddx, int(accumulate), stream_ptr())


def head_wgrad(x, dlogits, dw, db, N, V, Cc, K, ldx, scratch: "NormScratch" = None, innorm: Optional["InNorm"] = None):
    _f16(x, "head.x"); _f16(dlogits, "head.dlogits"); _f32(dw, "head.dw"); _f32(db, "head.db")
    if innorm is not None:
        assert innorm.c0 == 0 and scratch is not None and scratch.capacity * 2 >= 8 * (Cc + 1)
        call("nnz_seg_head_wgrad_innorm", ptr(x), ptr(innorm.tab), innorm.slope, ptr(dlogits), ptr(dw), ptr(db), N, V, Cc, K,
             ldx, ptr(scratch.acc), ptr(scratch.counter), stream_ptr())
        return
    if scratch is None:
        call("nnz_seg_head_wgrad", ptr(x), ptr(dlogits), ptr(dw), ptr(db), N, V, Cc, K, ldx, stream_ptr())
    else:
        assert scratch.capacity * 2 >= 8 * (Cc + 1)
        call("nnz_seg_head_wgrad_det", ptr(x), ptr(dlogits), ptr(dw), ptr(db), N, V, Cc, K, ldx, ptr(scratch.acc),
             ptr(scratch.counter), stream_ptr())


def instnorm_stats(x, stats, N, V, Cc, ldx, pre_zeroed: bool = False):
    _f16(x, "in.x"); _f32(stats, "in.stats")
    call("nnz_instnorm_stats", ptr(x), ptr(stats), N, V, Cc, ldx, int(pre_zeroed), stream_ptr())


def instnorm_lrelu_apply(x, stats, gamma, beta, y, N, V, Cc, ldx, ldy, eps, slope):
    _f16(x, "in.x"); _f16(y, "in.y"); _f32(stats, "in.stats"); _f32(gamma, "in.gamma"); _f32(beta, "in.beta")
    call("nnz_instnorm_lrelu_apply", ptr(x), ptr(stats), ptr(gamma), ptr(beta), ptr(y), N, V, Cc, ldx, ldy,
         eps, slope, stream_ptr())


def instnorm_lrelu_bwd(x, g, stats, gamma, beta, red, dx, N, V, Cc, ldx, ldg, lddx, eps, slope,
                       pre_zeroed: bool = False, dgamma=None, dbeta=None):
    _f16(x, "in.x"); _f16(g, "in.g"); _f16(dx, "in.dx"); _f32(stats, "in.stats"); _f32(red, "in.red")
    _f32(dgamma, "in.dgamma"); _f32(dbeta, "in.dbeta")
    call("nnz_instnorm_lrelu_bwd_reduce", ptr(x), ptr(g), ptr(stats), ptr(gamma), ptr(beta), ptr(red), N, V, Cc,
         ldx, ldg, eps, slope, int(pre_zeroed), stream_ptr())
    call("nnz_instnorm_lrelu_bwd_apply", ptr(x), ptr(g), ptr(stats), ptr(red), ptr(gamma), ptr(beta), ptr(dx), N, V,
         Cc, ldx, ldg, lddx, eps, slope, ptr(dgamma), ptr(dbeta), stream_ptr())


def instnorm_stats_det(x, N, V, Cc, ldx, scratch: NormScratch, gamma=None, beta=None, eps: float = 0.0, nstat=None,
                       sums=None):
    """deterministic statistics pass: nstat [N, C, 4] (needs gamma, beta, eps) and / or sums [N, C, 2] = {sum, sumsq}"""
    _f16(x, "in.x"); _f32(nstat, "in.nstat"); _f32(sums, "in.sums"); _f32(gamma, "in.gamma"); _f32(beta, "in.beta")
    assert N * Cc <= scratch.capacity
    call("nnz_instnorm_stats_det", ptr(x), N, V, Cc, ldx, ptr(scratch.acc), ptr(scratch.counter), ptr(gamma), ptr(beta),
         float(eps), ptr(nstat), ptr(sums), stream_ptr())


def instnorm_lrelu_apply_tab(x, nstat, y, N, V, Cc, ldx, ldy, slope):
    _f16(x, "in.x"); _f16(y, "in.y"); _f32(nstat, "in.nstat")
    call("nnz_instnorm_lrelu_apply_tab", ptr(x), ptr(nstat), ptr(y), N, V, Cc, ldx, ldy, float(slope), stream_ptr())


def instnorm_lrelu_bwd_tab(x, g, nstat, scratch: NormScratch, nred, dx, N, V, Cc, ldx, ldg, lddx, slope, dgamma=None,
                           dbeta=None):
    _f16(x, "in.x"); _f16(g, "in.g"); _f16(dx, "in.dx"); _f32(nstat, "in.nstat"); _f32(nred, "in.nred")
    _f32(dgamma, "in.dgamma"); _f32(dbeta, "in.dbeta")
    assert N * Cc <= scratch.capacity
    call("nnz_instnorm_lrelu_bwd_tab", ptr(x), ptr(g), ptr(nstat), ptr(scratch.acc), ptr(scratch.counter), ptr(nred),
         ptr(dx), N, V, Cc, ldx, ldg, lddx, float(slope), ptr(dgamma), ptr(dbeta), stream_ptr())


def instnorm_lrelu_bwd_apply_tab(x, g, nstat, nred, dx, N, V, Cc, ldx, ldg, lddx, slope):
    """the apply launch of the InstanceNorm + LeakyReLU backward alone: nred came from conv_tap_dgrad_normred"""
    _f16(x, "in.x"); _f16(g, "in.g"); _f16(dx, "in.dx"); _f32(nstat, "in.nstat"); _f32(nred, "in.nred")
    call("nnz_instnorm_lrelu_bwd_apply_tab", ptr(x), ptr(g), ptr(nstat), ptr(nred), ptr(dx), N, V, Cc, ldx, ldg, lddx,
         float(slope), stream_ptr())


def conv_tap_dgrad_normred(pt: PreparedTable, dy: torch.Tensor, w_packed: torch.Tensor, dx: torch.Tensor,
                           x_raw: torch.Tensor, ld_x: int, nstat: torch.Tensor, slope: float, scratch: NormScratch,
                           nred: torch.Tensor, dgamma: Optional[torch.Tensor], dbeta: Optional[torch.Tensor]) -> None:
    """data-gradient convolution whose epilogue also forms the norm-backward reductions of the layer below (the layer whose
    activation gradient `dx` is): nred [N, C, 2], dgamma / dbeta [C]"""
    _f16(dy, "conv.in"); _f16(dx, "conv.out"); _f16(w_packed, "conv.w"); _f16(x_raw, "conv.x_raw")
    _f32(nstat, "conv.nstat"); _f32(nred, "conv.nred"); _f32(dgamma, "conv.dgamma"); _f32(dbeta, "conv.dbeta")
    assert pt.table.N * pt.table.Cout <= scratch.capacity
    TIMER.wrap("conv_box_kernel", pt.flops,
               lambda: call("nnz_conv_tap_dgrad_normred", ptr(dy), ptr(dx), ptr(w_packed), C.byref(pt.desc), ptr(x_raw),
                            int(ld_x), ptr(nstat), float(slope), ptr(scratch.acc), ptr(scratch.counter), ptr(nred),
                            ptr(dgamma), ptr(dbeta), stream_ptr()))


def _logits_kind(t: torch.Tensor) -> int:
    if not t.is_cuda or t.dtype not in (torch.float16, torch.float32):
        raise _lib.HipCallError(f"loss: logits must be fp16/fp32 device tensors, got {t.dtype} on {t.device}")
    return 1 if t.dtype == torch.float16 else 0


NO_IGNORE = -32768


def dc_ce_forward(logits, target_i16, sums, B, Cc, V, ignore_label: int = NO_IGNORE):
    """sums are formed deterministically (fixed-point cross-workgroup adds, csrc/common.hpp): bit-identical run to run"""
    if target_i16.dtype != torch.int16 or not target_i16.is_cuda:
        raise _lib.HipCallError("loss: target must be an int16 device tensor")
    _f32(sums, "loss.sums")
    sc = det_scratch(logits.device, B * (3 * Cc + 1))
    call("nnz_dc_ce_loss_forward_det", ptr(logits), _logits_kind(logits), ptr(target_i16), ptr(sums), B, Cc, V,
         int(ignore_label), ptr(sc.acc), ptr(sc.counter), stream_ptr())


def dc_ce_backward(logits, target_i16, coef, dlogits, B, Cc, V, ignore_label: int = NO_IGNORE):
    _f32(coef, "loss.coef")
    if dlogits.dtype != logits.dtype:
        raise _lib.HipCallError("loss: dlogits dtype must equal logits dtype")
    call("nnz_dc_ce_loss_backward", ptr(logits), _logits_kind(logits), ptr(target_i16), ptr(coef), ptr(dlogits), B, Cc,
         V, int(ignore_label), stream_ptr())


def dc_ce_finalize(sums, loss_accum, coef, B, Cc, V, batch_dice, do_bg, smooth, weight_ce, weight_dice, ds_weight,
                   use_valid_count):
    _f32(sums, "loss.sums"), _f32(loss_accum, "loss.loss_accum"), _f32(coef, "loss.coef")
    call("nnz_dc_ce_loss_finalize", ptr(sums), ptr(loss_accum), ptr(coef), B, Cc, V, int(batch_dice), int(do_bg),
         float(smooth), float(weight_ce), float(weight_dice), float(ds_weight), int(use_valid_count), stream_ptr())


def dc_ce_backward_scaled(logits, target_i16, coef, gmul, dlogits, B, Cc, V, ignore_label: int = NO_IGNORE):
    _f32(coef, "loss.coef"), _f32(gmul, "loss.gmul")
    if dlogits.dtype != logits.dtype:
        raise _lib.HipCallError("loss: dlogits dtype must equal logits dtype")
    call("nnz_dc_ce_loss_backward_scaled", ptr(logits), _logits_kind(logits), ptr(target_i16), ptr(coef), ptr(gmul),
         ptr(dlogits), B, Cc, V, int(ignore_label), stream_ptr())


def argmax_tp_fp_fn(logits: torch.Tensor, target_i16: torch.Tensor, ignore_label: int = -32768):
    """(B, C, *spatial) logits + (B, 1, *spatial) int16 labels -> exact int64 (tp, fp, fn) per class, one pass"""
    if target_i16.dtype != torch.int16 or not target_i16.is_cuda:
        raise _lib.HipCallError("argmax_tp_fp_fn: target must be an int16 device tensor")
    logits = logits.contiguous()
    B, Cc = logits.shape[:2]
    V = logits[0, 0].numel()
    counts = torch.empty((Cc, 3), dtype=torch.int64, device=logits.device)
    call("nnz_argmax_tp_fp_fn", ptr(logits), _logits_kind(logits), ptr(target_i16.contiguous()), ptr(counts), B, Cc, V,
         int(ignore_label), stream_ptr())
    return counts[:, 0], counts[:, 1], counts[:, 2]


def _regions_i16(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.HipCallError("region targets must be device tensors")
    return (t if t.dtype == torch.int16 else t.to(torch.int16)).contiguous()


def dc_bce_forward(logits, target_regions_i16, sums, B, Cc, Ct, V):
    _f32(sums, "loss.sums")
    call("nnz_dc_bce_loss_forward", ptr(logits), _logits_kind(logits), ptr(target_regions_i16), ptr(sums), B, Cc, Ct, V,
         stream_ptr())


def dc_bce_backward(logits, target_regions_i16, coef, dlogits, B, Cc, Ct, V):
    _f32(coef, "loss.coef")
    call("nnz_dc_bce_loss_backward", ptr(logits), _logits_kind(logits), ptr(target_regions_i16), ptr(coef), ptr(dlogits),
         B, Cc, Ct, V, stream_ptr())


def region_tp_fp_fn(logits: torch.Tensor, target_regions: torch.Tensor):
    """(B, C, *spatial) logits + (B, C[+1], *spatial) 0/1 region targets (last channel = ignore mask if present) ->
    exact int64 (tp, fp, fn) per region for the prediction sigmoid(z) > 0.5"""
    logits = logits.contiguous()
    tgt = _regions_i16(target_regions)
    B, Cc = logits.shape[:2]
    V = logits[0, 0].numel()
    counts = torch.empty((Cc, 3), dtype=torch.int64, device=logits.device)
    call("nnz_region_tp_fp_fn", ptr(logits), _logits_kind(logits), ptr(tgt), ptr(counts), B, Cc, int(tgt.shape[1]), V,
         stream_ptr())
    return counts[:, 0], counts[:, 1], counts[:, 2]
