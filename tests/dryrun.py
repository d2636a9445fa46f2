"""Test helper: run the PlainConvUNet forward / backward SCHEDULE on CPU tensors with every kernel launch replaced by a
host stand-in.  Nothing is computed - the stand-ins only fill the parameter-gradient buffers they are handed with a
constant - so that schedule-level properties (gradient arena layout, reducer hand-over points, which parameters receive
a gradient) can be tested without a GPU.  Product code never imports this."""
import contextlib

import torch

from nnuzoo_amd.nets import plain_conv_unet as pcu


class _NoPack:
    def __init__(self, device):
        self.jobs = []

    def add(self, *a, **k):
        self.jobs.append(a)

    def run(self):
        pass


@contextlib.contextmanager
def stub_kernel_launches(fill: float = 1.0):
    ops = pcu.ops
    saved = {}

    def put(name, fn):
        saved[name] = getattr(ops, name)
        setattr(ops, name, fn)

    def nop(*a, **k):
        return None

    def wgrad_to_grad(pt, boxed, plain, ws, grad, *a, **k):
        grad.fill_(fill)

    def stem_wgrad(x, dy, dw, *a, **k):
        dw.fill_(fill)

    def head_wgrad(x, g, gw, gb, *a, **k):
        gw.fill_(fill)
        gb.fill_(fill)

    def norm_bwd(x, g, nstat, scratch, nred, dx, *a, dgamma=None, dbeta=None, **k):
        dgamma.fill_(fill)
        dbeta.fill_(fill)

    def dgrad_normred(pt, dy, w_packed, dx, x_raw, ld_x, nstat, slope, scratch, nred, dgamma, dbeta, *a, **k):
        # the fused data-gradient launch also writes the affine gradients of the layer below
        if dgamma is not None:
            dgamma.fill_(fill)
            dbeta.fill_(fill)

    def stats(x, N, V, Cc, ldx, scratch, gamma=None, beta=None, eps=0.0, nstat=None, sums=None):
        if sums is not None:
            sums.zero_()
            sums[:, :, 0] = fill / N   # the transposed conv's bias gradient is the sum over samples of this column

    class _Scratch:
        def __init__(self, device, n_times_c):
            self.capacity = n_times_c

    try:
        for n in ("stem_forward", "conv_tap_forward", "conv_tap_forward_norm", "convT_forward", "convT_dgrad",
                  "instnorm_lrelu_apply_tab", "instnorm_lrelu_bwd_apply_tab", "head_forward", "head_dgrad"):
            put(n, nop)
        put("instnorm_stats_det", stats)
        put("NormScratch", _Scratch)
        put("conv_tap_wgrad_to_grad", wgrad_to_grad)
        put("stem_wgrad", stem_wgrad)
        put("head_wgrad", head_wgrad)
        put("instnorm_lrelu_bwd_tab", norm_bwd)
        put("conv_tap_dgrad_normred", dgrad_normred)
        put("PackJobTable", _NoPack)
        put("DualPackTable", _NoPack)
        yield
    finally:
        for n, fn in saved.items():
            setattr(ops, n, fn)
