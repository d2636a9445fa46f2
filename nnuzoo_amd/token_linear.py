"""`TokenLinear`: nn.Linear (same parameters, state_dict, autocast numerics) for token-major activations with very many
tokens and few features - the in_proj / out_proj / patch merge / expand layers of the VSS blocks at 512^2 (524 288 tokens
x 16..256 features; reference: nets/m2net.py:97,103,258,300).

Under the fp16 autocast step the three products run on the hand-written kernels of csrc/token_linear.hip (one pass over
the activations, weights staged once per workgroup from the fp32 master parameter - no weight cast launch): forward and
input gradient are one MFMA kernel, the weight / bias gradient an fp32 register-tile kernel.  Shapes those kernels do not
serve (features beyond 256, fp32 steps) keep the library GEMMs; there the WEIGHT gradient dW = dY^T X - a few thousand
entries contracted over 10^5..10^6 tokens, for which the library picks a 16x16 macro-tile without split-K - is cut along
the token axis into one batched GEMM over strided views plus a small sum."""
from __future__ import annotations

import os

import ctypes as C

import torch

from . import backends as _backends
import torch.nn.functional as F
from torch import nn

from . import _lib
from ._lib import call, ptr, stream_ptr

MIN_TOKENS = 65536
H16_IO = os.environ.get("NNZ_TL_H16", "1") != "0"    # fp16 in / out on the fp32 matrix-core kernels (A/B: 0 = fp32 copies around them)
HIP_MIN_TOKENS = int(os.environ.get("NNZ_TL_MIN_TOKENS", "1024"))   # below this the call is launch-bound either way (A/B: env)
USE_HIP_KERNELS = os.environ.get("NNZ_TOKEN_LINEAR", "1") != "0"   # A/B switch for measurements
MAX_FEATURES = 256
SMALL_F32 = os.environ.get("NNZ_TL_SMALL_F32", "1") != "0"   # autocast calls outside the fp16 kernel's range: fp32 MFMA kernels, not the library


def _chunks(T: int) -> int:
    nc = 1
    while nc < 128 and T % (nc * 2) == 0 and T // (nc * 2) >= 4096:
        nc *= 2
    return nc


def _wgrad_chunks(T: int, fin: int, fout: int) -> int:
    """how many token chunks the weight-gradient GEMM dW[fout, fin] = dY^T X is cut into (a batched GEMM + a sum over the
    chunks).  Two regimes make one library GEMM slow: very many tokens (`_chunks`), and a SMALL result - a 256 x 256 dW is one
    256 x 256 macro tile = one workgroup walking all T tokens (SwT2Net: 20 such calls at 205 us and 8 at 225 us per step,
    tools/probes/swt_slow_conv_probe.py); cutting T gives the chip something to do."""
    nc = _chunks(T)
    wgs = -(-fout // 256) * -(-fin // 256)
    if wgs < 32 and T >= 256:
        want = min(64 // wgs, T // 64)
        for c in range(want, 1, -1):
            if T % c == 0:
                nc = max(nc, c)
                break
    return nc


def _hip_ok(kr: int, mo: int) -> bool:
    return bool(_lib.load().nnz_token_linear_supported(int(kr), int(mo)))


class _HipTokenLinearFn(torch.autograd.Function):
    """fp16 autocast path on csrc/token_linear.hip.  x: (..., K) fp16 contiguous device tensor; weight / bias fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        N, K = weight.shape
        T = x.numel() // K
        y = torch.empty((*x.shape[:-1], N), dtype=torch.float16, device=x.device)
        call("nnz_token_linear_forward", ptr(x), ptr(weight), ptr(bias), ptr(y), T, K, N, 0, stream_ptr())
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)             # the parameter objects themselves: the deferred path sets their .grad
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        N, K = weight.shape
        T = x.numel() // K
        dy = dy if dy.dtype == torch.float16 else dy.to(torch.float16)
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if _hip_ok(N, K):
                dx = torch.empty_like(x)
                call("nnz_token_linear_forward", ptr(dy), ptr(weight), None, ptr(dx), T, N, K, 1, stream_ptr())
            else:
                dx = (dy.reshape(-1, N) @ weight.to(torch.float16)).view(x.shape)
        need_w, need_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if need_w or need_b:
            wp, bp = ctx.params
            if N <= 256 and K <= 256 and N % 8 == 0 and K % 8 == 0 and GROUP_TL_WGRAD and _DEFER["on"] and need_w and wp.is_leaf \
                    and (bp is None or bp.is_leaf) and not _has_grad_hooks(wp) and not (bp is not None and _has_grad_hooks(bp)):
                # queued: ONE grouped launch + one fold launch at the end of the backward pass (csrc/token_linear.hip
                # tl_wgrad_group_kernel) computes and assigns the gradients
                _DEFER["tl_jobs"].append((dy, x, wp, bp if need_b else None))
                return dx, None, None
            if (N // 8) * (K // 8) <= 256:
                if TWO_STAGE:
                    # per-workgroup partial blocks + a fixed-order fold (csrc/common.hpp fold_partials): bit-reproducible, and
                    # no zero-fill launch
                    buf = torch.empty(N * K + N, dtype=torch.float32, device=x.device)
                    dw, dbv = buf[:N * K].view(N, K), buf[N * K:]
                    nws = int(_lib.load().nnz_token_linear_wgrad_workspace_floats(T, N, K))
                    ws = torch.empty(nws, dtype=torch.float32, device=x.device)
                    call("nnz_token_linear_wgrad_ws", ptr(dy), ptr(x), ptr(dw), ptr(dbv) if need_b else None, ptr(ws), nws,
                         T, N, K, stream_ptr())
                else:
                    buf = torch.zeros(N * K + N, dtype=torch.float32, device=x.device)
                    dw, dbv = buf[:N * K].view(N, K), buf[N * K:]
                    call("nnz_token_linear_wgrad", ptr(dy), ptr(x), ptr(dw), ptr(dbv) if need_b else None, T, N, K,
                         stream_ptr())
                db = dbv if need_b else None
            else:
                dy2, x2 = dy.reshape(-1, N), x.reshape(-1, K)
                nc = _chunks(T)
                if nc > 1:
                    dw = torch.bmm(dy2.view(nc, T // nc, N).transpose(1, 2), x2.view(nc, T // nc, K)).sum(
                        0, dtype=torch.float32)
                else:
                    dw = (dy2.t() @ x2).float()
                db = dy2.sum(0, dtype=torch.float32) if need_b else None
        return dx, dw, db


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
                nc = _wgrad_chunks(T, fin, fout)
                if nc > 1:
                    part = torch.bmm(dy2.view(nc, T // nc, fout).transpose(1, 2), x2.view(nc, T // nc, fin))
                    dw = part.sum(0, dtype=torch.float32).to(wdt)
                else:
                    dw = (dy2.t() @ x2).to(wdt)
            if has_bias and ctx.needs_input_grad[2]:
                db = dy2.sum(0, dtype=torch.float32).to(bdt)
        return dx, dw, db


USE_DENSE32 = os.environ.get("NNZ_DENSE32", "1") != "0"   # A/B switch: fp32 MFMA kernels (csrc/dense32.hip) vs GEMM library
DENSE32_MIN_TOKENS = 64
_WS = {}


def _d32_workspace(device, floats: int) -> torch.Tensor:
    """partial-sum workspace of the fp32 weight gradients, per (device, stream): calls on one stream use it one after the other,
    two streams never share one (see swin_block._workspace)"""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < floats:
        ws = torch.empty(max(floats, 1 << 22), dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


def dense32_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """fp32 step (no autocast), device tensors, features multiples of 4: the hand-written fp32 MFMA path"""
    if not (USE_DENSE32 and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32):
        return False
    if torch.is_autocast_enabled():
        return False
    N, K = weight.shape
    return K % 4 == 0 and N % 4 == 0 and x.numel() // K >= DENSE32_MIN_TOKENS


# ---- deferred, grouped weight gradients ------------------------------------------------------------------------------------
# Inside `with deferred_wgrads():` (the trainers wrap loss.backward() in it) the fp32 Linear nodes do NOT launch their
# weight-gradient kernels: they queue (dy, x, parameter) and return None for the parameter gradients; leaving the context
# runs every queued problem in ONE grouped launch (+ one fold launch, csrc/dense32.hip dense32_group_*) and assigns /
# accumulates `.grad` itself.  A Swin step holds ~670 such problems of 18-30 us each (21 ms of a 104 ms SwT2Net step); they
# do not sit on the data-gradient chain.  Results are bit-identical to the per-layer launches (same split rule, same
# arithmetic, fixed fold order).  Outside the context (torch.autograd.grad, parity tests) nothing changes.
GROUP_WGRAD = os.environ.get("NNZ_DENSE32_GROUP", "1") != "0"
# weight gradients whose token ranges are spread over many workgroups (token Linear outside the grouped launches, x_proj, depthwise
# conv + SiLU): partial blocks + fixed-order fold, bit-reproducible (tests/test_two_stage_wgrads_gpu.py).  Default since the end of
# round 6: with the grouped launches carrying most of these gradients the two-stage form costs nothing any more (same-box A/B at 512^2,
# batch 2: M2Net 73.77 / 73.86 ms with it, 73.83 / 73.91 without; M2NetP 56.16 vs 56.23) and it is what makes the M2Net training step
# bit-identical run to run in the DEFAULT mode (tests/test_zoo_determinism_gpu.py).  NNZ_TWO_STAGE_WGRADS=0: fp32 atomics into a
# zero-filled gradient (rounds 2-5; 0.9 - 1.4 % faster in round 4, before the grouped launches).
TWO_STAGE = os.environ.get("NNZ_TWO_STAGE_WGRADS", "1") != "0"
_DEFER = {"on": False, "jobs": [], "folds": [], "tl_jobs": [], "xp_jobs": []}
GROUP_TL_WGRAD = os.environ.get("NNZ_TL_GROUP", "1") != "0"     # fp16 token Linears: weight gradients grouped like the fp32 ones
_GROUP_KEEP = []          # host tables captured into a hipGraph must outlive it
_HOST_CACHE, _HOST_EVENTS = {}, {}


def _has_grad_hooks(p: torch.Tensor) -> bool:
    """tensor hooks / post-accumulate-grad hooks (DDP-style reducers, user hooks) only see gradients that autograd delivers: a
    parameter that carries any keeps its weight gradient on the ordinary path"""
    return bool(getattr(p, "_backward_hooks", None)) or bool(getattr(p, "_post_accumulate_grad_hooks", None))


class deferred_wgrads:
    """Valid around `.backward()` only: the queued Linear nodes return None for their weight / bias gradients and `.grad` is
    assigned by the grouped launch at exit - `torch.autograd.grad(..., inputs=params)` inside the context gets nothing for them.
    Every queued (dy, x) pair stays alive until the exit.  Process-global, not thread-local: one backward at a time."""

    def __enter__(self):
        self._outer = _DEFER["on"]
        _DEFER["on"] = GROUP_WGRAD or self._outer
        return self

    def __exit__(self, et, ev, tb):
        if not self._outer:
            _DEFER["on"] = False
            jobs, _DEFER["jobs"] = _DEFER["jobs"], []
            folds, _DEFER["folds"] = _DEFER["folds"], []
            tl_jobs, _DEFER["tl_jobs"] = _DEFER["tl_jobs"], []
            xp_jobs, _DEFER["xp_jobs"] = _DEFER["xp_jobs"], []
            if et is None and tl_jobs:
                _flush_table_group(_TlKind, tl_jobs)
            if et is None and xp_jobs:
                _flush_table_group(_XpKind, xp_jobs)
            if et is None and jobs:
                # one grouped launch per tile class (64 x 64 / 128 x 128 tiles; csrc/dense32.hip d32_group_class); the fold-only
                # records (LayerNorm dgamma | dbeta partials of the fused Swin blocks) ride in the first one
                lib = _lib.load()
                # (bit 1 of the class: fp16 dy / x - the operand type is a template parameter of the kernel)
                def _cls(j):
                    return int(lib.nnz_dense32_group_class(j[1].shape[0], j[2].shape[1], j[2].shape[0])) | \
                        (2 if j[0].dtype == torch.float16 else 0)
                for cls in (0, 1, 2, 3):
                    sub = [j for j in jobs if _cls(j) == cls]
                    if sub:
                        _flush_group(sub, cls, folds)
                        folds = []
            if et is None and folds:
                for part, n, parts, assign in folds:        # no Linear job in this pass to ride with
                    assign(part.view(parts, n).sum(0))
        return False


def defer_fold(part: torch.Tensor, n: int, parts: int, assign) -> bool:
    """queue dst[i] = sum_q part[q][i] (q < parts, i < n) for the grouped launch at the end of the backward pass; `assign(dst)` is
    called with the [n] result tensor right after the launch has been issued.  False outside deferred_wgrads()."""
    if not _DEFER["on"]:
        return False
    _DEFER["folds"].append((part, n, parts, assign))
    return True


def _host_table(key, nbytes: int):
    """pinned host buffer for a job table.  Pinned memory cannot be allocated while a stream is capturing: the eager warm-up passes
    that precede every capture leave their buffer in _HOST_CACHE under the pass's shape signature; a capturing flush TAKES it (the
    captured copy node re-reads it at every replay, so no later flush may write to it)."""
    capturing = torch.cuda.is_current_stream_capturing()
    host = _HOST_CACHE.pop(key, None)
    ev = _HOST_EVENTS.pop(key, None)
    if host is not None and ev is not None and not capturing:
        ev.synchronize()                  # the previous copy out of this buffer has run
    if host is None or host.numel() != nbytes:
        if capturing:
            raise RuntimeError("grouped weight gradients: no pinned table from a warm-up pass for this capture - run one "
                               "eager pass with the same shapes before capturing")
        host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    return host, capturing


def _table_to_device(host, key, capturing, dev):
    tab = torch.empty(host.numel(), dtype=torch.uint8, device=dev)
    tab.copy_(host, non_blocking=True)
    if capturing:
        _GROUP_KEEP.append(host)          # the captured copy node reads this buffer at every replay
    else:
        ev = torch.cuda.Event()
        ev.record()
        _HOST_CACHE[key] = host
        _HOST_EVENTS[key] = ev
    return tab


def _accumulate(p, g):
    if p.grad is None:
        p.grad = g
    else:
        p.grad.add_(g)


class _TlKind:
    """fp16 token Linear weight gradients: job = (dy f16 [T][N], x f16 [T][K], weight, bias or None)"""
    name, rec, launch = "tl", "nnz_token_linear_wgrad_group_record_bytes", "nnz_token_linear_wgrad_group_launch"

    @staticmethod
    def sig(j):
        return (j[1].numel() // j[2].shape[1],) + tuple(j[2].shape) + (j[3] is not None,)

    @staticmethod
    def plan(j, wgs, lds, wsf):
        N, K = j[2].shape
        call("nnz_token_linear_wgrad_group_plan", j[1].numel() // K, N, K, C.addressof(wgs), C.addressof(lds), C.addressof(wsf))
        return N * K + N, wsf.value // (N * K + N)       # floats of the folded result, partial blocks (one per token range)

    @staticmethod
    def fill(rec_ptr, j, part_ptr, wg0):
        N, K = j[2].shape
        call("nnz_token_linear_wgrad_group_fill", rec_ptr, ptr(j[0]), ptr(j[1]), part_ptr, j[1].numel() // K, N, K, wg0)

    @staticmethod
    def assign(j, dst):
        N, K = j[2].shape
        _accumulate(j[2], dst[:N * K].view(N, K))
        if j[3] is not None:
            _accumulate(j[3], dst[N * K:])


class _XpKind:
    """SS2D x_proj weight gradients: job = (dP [2][B][C2][L], x2 [2][B][Di][L], x_proj_weight parameter [4][Cp][Di])"""
    name, rec, launch = "xp", "nnz_ss2d_xproj_backward_w_group_record_bytes", "nnz_ss2d_xproj_backward_w_group_launch"

    @staticmethod
    def sig(j):
        return tuple(j[0].shape) + tuple(j[1].shape)

    @staticmethod
    def plan(j, wgs, lds, wsf):
        _, B, C2, L = j[0].shape
        Di = j[1].shape[2]
        call("nnz_ss2d_xproj_backward_w_group_plan", B, Di, C2, L, C.addressof(wgs), C.addressof(lds), C.addressof(wsf))
        return 2 * C2 * Di, wsf.value // (2 * C2 * Di)     # partial matrices: one per token range

    @staticmethod
    def fill(rec_ptr, j, part_ptr, wg0):
        _, B, C2, L = j[0].shape
        Di = j[1].shape[2]
        call("nnz_ss2d_xproj_backward_w_group_fill", rec_ptr, ptr(j[0]), ptr(j[1]), part_ptr, B, Di, C2, L, C2 // 2, wg0)

    @staticmethod
    def assign(j, dst):
        _accumulate(j[2], dst.view(j[2].shape))


def _flush_table_group(kind, jobs) -> None:
    """one grouped launch (per-workgroup partial blocks into a workspace) + one fold launch for the queued jobs of one kernel
    family; gradients are assigned / accumulated here.  Bit-identical from pass to pass (fixed partition, fixed fold order)."""
    import numpy as np
    lib = _lib.load()
    dev = jobs[0][0].device
    rb, rbf = int(getattr(lib, kind.rec)()), int(lib.nnz_dense32_group_record_bytes(1))
    wgs, lds, wsf = C.c_int(0), C.c_int(0), C.c_long(0)
    plans = []
    for j in jobs:
        n_out, parts = kind.plan(j, wgs, lds, wsf)
        plans.append((wgs.value, lds.value, wsf.value, n_out, parts, (n_out + 255) // 256))
    total_wgs, total_blks = sum(p[0] for p in plans), sum(p[5] for p in plans)
    ws = torch.empty(sum(p[2] for p in plans), dtype=torch.float32, device=dev)
    o_fold = (len(jobs) * rb + 15) // 16 * 16
    o_wg = (o_fold + len(jobs) * rbf + 15) // 16 * 16
    o_blk = o_wg + total_wgs * 4
    nbytes = o_blk + total_blks * 4
    key = (kind.name,) + tuple(kind.sig(j) for j in jobs)
    host, capturing = _host_table(key, nbytes)
    hp, hn = host.data_ptr(), host.numpy()
    wg_job = hn[o_wg:o_wg + total_wgs * 4].view(np.int32)
    blk_job = hn[o_blk:o_blk + total_blks * 4].view(np.int32)
    wg0 = blk0 = ws_off = 0
    outs = []
    for i, (j, (nw, _, nf, n_out, parts, nb)) in enumerate(zip(jobs, plans)):
        part = ws.data_ptr() + 4 * ws_off
        dst = torch.empty(n_out, dtype=torch.float32, device=dev)
        kind.fill(hp + i * rb, j, part, wg0)
        call("nnz_dense32_group_fill_fold", hp + o_fold + i * rbf, part, ptr(dst), n_out, parts, blk0)
        wg_job[wg0:wg0 + nw] = i
        blk_job[blk0:blk0 + nb] = i
        wg0, blk0, ws_off = wg0 + nw, blk0 + nb, ws_off + nf
        outs.append((j, dst))
    tab = _table_to_device(host, key, capturing, dev)
    base = tab.data_ptr()
    call(kind.launch, base, base + o_wg, total_wgs, max(p[1] for p in plans), stream_ptr())
    call("nnz_group_fold_launch", base + o_fold, base + o_blk, total_blks, stream_ptr())
    for j, dst in outs:
        kind.assign(j, dst)


def defer_xproj_wgrad(dP, x2, param) -> bool:
    """queue the x_proj weight gradient of one SS2D block for the pass's grouped launch (False outside deferred_wgrads() or when the
    parameter is not a plain leaf)"""
    if not (GROUP_TL_WGRAD and _DEFER["on"] and isinstance(param, torch.Tensor) and param.is_leaf and param.requires_grad
            and param.dtype == torch.float32 and not _has_grad_hooks(param)):
        return False
    _DEFER["xp_jobs"].append((dP, x2, param))
    return True


def _flush_group(jobs, tile_class: int = 0, folds=()) -> None:
    """jobs: (dy2, x2, weight, bias or None[, (dp_rand, keep, rows_per_sample, samples)]) - the optional fifth entry scales the dy
    operand per token (weight gradient of a DropPath branch, csrc/dense32.hip nnz_dense32_group_fill_scaled)"""
    import numpy as np
    lib = _lib.load()
    dev = jobs[0][0].device
    rb_job, rb_fold = int(lib.nnz_dense32_group_record_bytes(0)), int(lib.nnz_dense32_group_record_bytes(1))
    plans = []
    wgs, blks, ws_f = C.c_int(0), C.c_int(0), C.c_long(0)
    for job in jobs:
        dy2, x2, w, b = job[:4]
        N, K = w.shape
        call("nnz_dense32_group_plan", x2.shape[0], K, N, C.addressof(wgs), C.addressof(blks), C.addressof(ws_f))
        plans.append((wgs.value, blks.value, ws_f.value))
    fold_blks = [(n + 255) // 256 for _, n, _, _ in folds]
    nfold = sum(1 for p in plans if p[1] > 0) + len(folds)
    total_wgs, total_blks = sum(p[0] for p in plans), sum(p[1] for p in plans) + sum(fold_blks)
    ws = torch.empty(max(1, sum(p[2] for p in plans)), dtype=torch.float32, device=dev)
    # one pinned host buffer: [job records | fold records | workgroup -> job | fold block -> fold job]
    o_fold = (len(jobs) * rb_job + 15) // 16 * 16
    o_wg = (o_fold + nfold * rb_fold + 15) // 16 * 16
    o_blk = o_wg + total_wgs * 4
    # Pinned memory cannot be allocated while a stream is capturing (it invalidates the capture): the eager warm-up passes
    # that precede every capture leave their buffer in _HOST_CACHE under the pass's shape signature; a capturing flush TAKES
    # it (the captured copy node re-reads it at every replay, so no later flush may write to it).
    nbytes = o_blk + max(1, total_blks) * 4
    key = (tile_class,) + tuple((j[1].shape[0],) + tuple(j[2].shape) + (j[3] is not None, len(j) > 4 and j[4] is not None,
                                                                         j[0].dtype == torch.float16)
                                for j in jobs) + tuple((n, parts) for _, n, parts, _ in folds)
    capturing = torch.cuda.is_current_stream_capturing()
    host = _HOST_CACHE.pop(key, None)
    ev = _HOST_EVENTS.pop(key, None)
    if host is not None and ev is not None and not capturing:
        ev.synchronize()                  # the previous copy out of this buffer has run
    if host is None or host.numel() != nbytes:
        if capturing:
            raise RuntimeError("grouped weight gradients: no pinned table from a warm-up pass for this capture - run one "
                               "eager pass with the same shapes before capturing")
        host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    hp = host.data_ptr()
    hn = host.numpy()
    wg_job = hn[o_wg:o_wg + total_wgs * 4].view(np.int32)
    blk_job = hn[o_blk:o_blk + total_blks * 4].view(np.int32)
    wg0 = blk0 = fi = 0
    ws_off = 0
    outs = []
    for j, (job, (nw, nb, nf)) in enumerate(zip(jobs, plans)):
        dy2, x2, w, b = job[:4]
        dp = job[4] if len(job) > 4 else None
        N, K = w.shape
        dw = torch.empty((N, K), dtype=torch.float32, device=dev)
        db = torch.empty(N, dtype=torch.float32, device=dev) if b is not None else None
        if dy2.dtype == torch.float16:
            assert dp is None and x2.dtype == torch.float16
            call("nnz_dense32_group_fill_h16", hp + j * rb_job, (hp + o_fold + fi * rb_fold) if nb else None, ptr(dy2), ptr(x2),
                 ptr(dw), ptr(db), (ws.data_ptr() + 4 * ws_off) if nf else None, x2.shape[0], K, N, wg0, blk0)
        elif dp is None:
            call("nnz_dense32_group_fill", hp + j * rb_job, (hp + o_fold + fi * rb_fold) if nb else None, ptr(dy2), ptr(x2),
                 ptr(dw), ptr(db), (ws.data_ptr() + 4 * ws_off) if nf else None, x2.shape[0], K, N, wg0, blk0)
        else:
            call("nnz_dense32_group_fill_scaled", hp + j * rb_job, (hp + o_fold + fi * rb_fold) if nb else None, ptr(dy2),
                 ptr(x2), ptr(dw), ptr(db), (ws.data_ptr() + 4 * ws_off) if nf else None, x2.shape[0], K, N, wg0, blk0,
                 ptr(dp[0]), float(dp[1]), int(dp[2]), int(dp[3]))
        wg_job[wg0:wg0 + nw] = j
        if nb:
            blk_job[blk0:blk0 + nb] = fi
            fi += 1
        wg0, blk0, ws_off = wg0 + nw, blk0 + nb, ws_off + nf
        outs.append((w, dw, b, db))
    fold_outs = []
    for (part, n, parts, assign), nb in zip(folds, fold_blks):
        dst = torch.empty(n, dtype=torch.float32, device=dev)
        call("nnz_dense32_group_fill_fold", hp + o_fold + fi * rb_fold, ptr(part), ptr(dst), n, parts, blk0)
        blk_job[blk0:blk0 + nb] = fi
        fi += 1
        blk0 += nb
        fold_outs.append((assign, dst))
    tab = torch.empty(host.numel(), dtype=torch.uint8, device=dev)
    tab.copy_(host, non_blocking=True)
    if capturing:
        _GROUP_KEEP.append(host)          # the captured copy node reads this buffer at every replay
    else:
        # the copy above is stream-ordered; the next flush with this signature rewrites the buffer only after synchronising
        # with it (same stream: its own copy is ordered behind this one; the host writes are guarded by the event below)
        ev = torch.cuda.Event()
        ev.record()
        _HOST_CACHE[key] = host
        _HOST_EVENTS[key] = ev
    base = tab.data_ptr()
    call("nnz_dense32_group_launch", base, base + o_wg, total_wgs, base + o_fold, base + o_blk, total_blks, int(tile_class),
         stream_ptr())
    for w, dw, b, db in outs:
        for p, g in ((w, dw), (b, db)):
            if g is None:
                continue
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
    for assign, dst in fold_outs:
        assign(dst)


def _d32_forward(x2, weight, bias, y, y_act, T, K, N, gelu):
    """y = x W^T + b on csrc/dense32.hip through the entry point that cuts skinny products along the contraction (split-K)"""
    from .swin_block import _workspace
    ws = _workspace(x2.device, int(_lib.load().nnz_dense32_splitk_workspace_floats(T, K, N)))
    if x2.dtype == torch.float16:          # fp16 activations in and out, fp32 weights and accumulation (no cast launches)
        assert y.dtype == torch.float16 and y_act is None and not gelu
        call("nnz_dense32_forward_h16", ptr(x2), ptr(weight), ptr(bias), ptr(y), T, K, N, ptr(ws), stream_ptr())
        return
    call("nnz_dense32_forward_fused", ptr(x2), ptr(weight), ptr(bias), ptr(y), ptr(y_act), T, K, N, int(gelu), None, None, 0.0, None,
         None, None, 0, 0, 0, 0, None, None, 1.0, 1, 1, ptr(ws), stream_ptr())


def _d32_dgrad(dy2, weight, h, dx, T, K, N):
    from .swin_block import _workspace
    ws = _workspace(dy2.device, int(_lib.load().nnz_dense32_splitk_workspace_floats(T, N, K)))
    if dy2.dtype == torch.float16:
        assert dx.dtype == torch.float16 and h is None
        call("nnz_dense32_dgrad_h16", ptr(dy2), ptr(weight), ptr(dx), T, K, N, ptr(ws), stream_ptr())
        return
    call("nnz_dense32_dgrad_fused", ptr(dy2), ptr(weight), ptr(h), ptr(dx), T, K, N, None, 1.0, 1, 1, ptr(ws), stream_ptr())


def _d32_backward_products(dy2, x2, weight, need_x, need_w, need_b, h=None, bias=None):
    """the three products of a Linear's backward on csrc/dense32.hip: dx (optionally times GELU'(h)), dW, db.  Inside
    deferred_wgrads() the weight / bias gradients are queued for the grouped launch and returned as None"""
    N, K = weight.shape
    T = x2.shape[0]
    dx = dw = db = None
    if need_x:
        dx = torch.empty((T, K), dtype=dy2.dtype, device=dy2.device)
        _d32_dgrad(dy2, weight, h, dx, T, K, N)
    if need_w or need_b:
        if _DEFER["on"] and need_w and weight.is_leaf and (bias is None or bias.is_leaf) and not _has_grad_hooks(weight) \
                and not (bias is not None and _has_grad_hooks(bias)):
            _DEFER["jobs"].append((dy2, x2, weight, bias if need_b else None))
            return dx, None, None
        if dy2.dtype == torch.float16:     # outside a deferred pass (module-level tests): the fp32 entry point on fp32 copies
            dy2, x2 = dy2.float(), x2.float()
        dw = torch.empty((N, K), dtype=torch.float32, device=dy2.device)
        db = torch.empty(N, dtype=torch.float32, device=dy2.device) if need_b else None
        ws = _d32_workspace(dy2.device, int(_lib.load().nnz_dense32_wgrad_workspace_floats(T, K, N)))
        call("nnz_dense32_wgrad", ptr(dy2), ptr(x2), ptr(dw), ptr(db), ptr(ws), T, K, N, stream_ptr())
    return dx, dw, db


class _Dense32LinearFn(torch.autograd.Function):
    """y = x W^T + b in fp32 on v_mfma_f32_32x32x2_f32 (csrc/dense32.hip): bias in the epilogue; backward = one input
    gradient launch + one weight/bias gradient launch (+ a fixed-order fold of its token splits) - no library GEMM, no
    reduce / fill / copy launches.  Reference numerics: torch fp32 F.linear (nnUNetTrainerSwT2Net.py:112-130, no autocast)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        N, K = weight.shape
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        T = x2.shape[0]
        y = torch.empty((T, N), dtype=x2.dtype, device=x.device)      # fp32, or fp16 in / fp16 out (autocast nets)
        _d32_forward(x2, weight, bias, y, None, T, K, N, 0)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)             # the parameter objects themselves (leaves): the deferred path sets their .grad
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        N, K = weight.shape
        dy2 = dy.reshape(-1, N)
        if not dy2.is_contiguous() or dy2.dtype != x2.dtype:
            dy2 = dy2.to(x2.dtype).contiguous()
        dx, dw, db = _d32_backward_products(dy2, x2, ctx.params[0], ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                            ctx.has_bias and ctx.needs_input_grad[2], bias=ctx.params[1])
        return (None if dx is None else dx.view(ctx.xshape)), (dw if ctx.needs_input_grad[1] else None), db


class _Dense32MlpFn(torch.autograd.Function):
    """fc2(GELU(fc1(x))) - the Mlp of the Swin / ViT blocks (swt2net.py:496-515, dropout 0) as four launches forward and
    backward each: GELU is applied in fc1's epilogue (which stores the pre-activation too), GELU' in the epilogue of fc2's
    input gradient; no element-wise launch touches the (tokens x 4C) hidden tensor."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        Hd, K = w1.shape
        N = w2.shape[0]
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        T = x2.shape[0]
        dev = x.device
        h = torch.empty((T, Hd), dtype=torch.float32, device=dev)
        a = torch.empty((T, Hd), dtype=torch.float32, device=dev)
        _d32_forward(x2, w1, b1, h, a, T, K, Hd, 1)
        y = torch.empty((T, N), dtype=torch.float32, device=dev)
        _d32_forward(a, w2, b2, y, None, T, Hd, N, 0)
        ctx.save_for_backward(x2, w1, w2, h, a)
        ctx.params = (w1, b1, w2, b2)
        ctx.bias = (b1 is not None, b2 is not None)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, h, a = ctx.saved_tensors
        N = w2.shape[0]
        dy2 = dy.reshape(-1, N)
        if not dy2.is_contiguous() or dy2.dtype != torch.float32:
            dy2 = dy2.float().contiguous()
        ni = ctx.needs_input_grad
        p = ctx.params
        dh, dw2, db2 = _d32_backward_products(dy2, a, p[2], True, ni[3], ctx.bias[1] and ni[4], h=h, bias=p[3])   # dh = (dy W2) * GELU'(h)
        dx, dw1, db1 = _d32_backward_products(dh, x2, p[0], ni[0], ni[1], ctx.bias[0] and ni[2], bias=p[1])
        return (None if dx is None else dx.view(ctx.xshape)), (dw1 if ni[1] else None), db1, (dw2 if ni[3] else None), db2


def mlp_gelu(x: torch.Tensor, fc1: nn.Linear, fc2: nn.Linear) -> torch.Tensor:
    """fc2(GELU(fc1(x))); the fused fp32 path when it applies, the module-by-module form otherwise"""
    if dense32_ok(x, fc1.weight) and dense32_ok(x, fc2.weight) and fc2.weight.shape[1] == fc1.weight.shape[0]:
        _backends.note(fc1, "hip-f32")
        _backends.note(fc2, "hip-f32")
        return _Dense32MlpFn.apply(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias)
    return fc2(F.gelu(fc1(x)))


class TokenLinear(nn.Linear):
    #: `self.backend` (set by the first call, nnuzoo_amd/backends.py): which implementation the most recent forward took -
    #: "hip-f16" / "hip-f32" / "library"; tests and the bench read it
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        tokens = x.numel() // max(1, x.shape[-1])
        if dense32_ok(x, self.weight):
            _backends.note(self, "hip-f32")
            return _Dense32LinearFn.apply(x, self.weight, self.bias)
        if USE_HIP_KERNELS and x.is_cuda and tokens >= HIP_MIN_TOKENS and x.is_contiguous() and self.weight.dtype == torch.float32 \
                and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.float16 \
                and x.dtype in (torch.float16, torch.float32) and self.in_features % 8 == 0 \
                and _hip_ok(self.in_features, self.out_features):
            xh = x if x.dtype == torch.float16 else x.to(torch.float16)
            _backends.note(self, "hip-f16")
            return _HipTokenLinearFn.apply(xh, self.weight, self.bias)
        # the reason is part of the record (tests assert that bench configurations only take this branch for small calls)
        if x.is_cuda and tokens < HIP_MIN_TOKENS and torch.is_autocast_enabled():
            why = "small"            # fewer than HIP_MIN_TOKENS tokens: launch-bound either way
        elif x.is_cuda and torch.is_autocast_enabled() and (self.in_features % 8 or self.in_features > MAX_FEATURES
                                                            or self.out_features > 4 * MAX_FEATURES
                                                            or not _hip_ok(self.in_features, self.out_features)):
            why = "features"         # feature counts outside the f16 token kernel's set
        else:
            why = "other"            # CPU tensor, dtype / layout outside both kernel families, switched off by environment
        # round 5: what the fp16 token kernel does not take under autocast - fewer than HIP_MIN_TOKENS tokens, feature counts
        # outside its set (the 8^2 ... 32^2 levels of the Mamba nets: 97 of M2Net's 272 Linear layers) - runs on the fp32 matrix-core
        # kernels (csrc/dense32.hip: fp32 operands from the fp32 master weights, split-K for the skinny shapes) instead of the GEMM
        # library: at least the precision of the reference's fp16-autocast GEMM, output rounded to fp16 like autocast's, no weight
        # cast launch, and its weight gradient joins the pass's grouped launch.  NNZ_TL_SMALL_F32=0 keeps the library path.
        if SMALL_F32 and why in ("small", "features") and x.is_cuda and torch.get_autocast_dtype("cuda") == torch.float16 \
                and self.weight.dtype == torch.float32 and self.in_features % 4 == 0 and self.out_features % 4 == 0 \
                and x.dtype in (torch.float16, torch.float32):
            _backends.note(self, "hip-f32", why=why)
            with torch.autocast("cuda", enabled=False):
                if x.dtype == torch.float16 and H16_IO:   # fp16 rows in, fp16 rows out: converted while staged / stored (no cast launches)
                    return _Dense32LinearFn.apply(x, self.weight, self.bias)
                y = _Dense32LinearFn.apply(x.float(), self.weight, self.bias)
            return y.to(torch.float16)
        self.backend_why = why
        _backends.note(self, "library", why=why)
        if x.is_cuda and x.is_contiguous() and x.dtype in (torch.float16, torch.float32) \
                and _wgrad_chunks(tokens, self.in_features, self.out_features) > 1:
            return _TallLinearFn.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)
