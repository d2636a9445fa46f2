"""per-operator times of the fp32 separable-conv path at SwT2Net's shapes (stems: B = 2, 512^2 / 4^k maps; RSU4F: 16^2, 8^2)
Usage (GPU box): python tools/probes/sepconv32_times.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnuzoo_amd.sepconv32 import _BnReluFn, _Dw3x3Fn, _Pointwise1x1Fn


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for B, H, C, N in [(2, 256, 64, 64), (2, 128, 128, 128), (2, 64, 256, 256), (2, 512, 32, 32), (2, 16, 512, 512), (2, 8, 1024, 512)]:
    x = torch.randn(B, H, H, C, device="cuda", requires_grad=True)
    w = torch.randn(C, 1, 3, 3, device="cuda", requires_grad=True)
    pw = torch.randn(N, C, 1, 1, device="cuda", requires_grad=True)
    g, b = torch.ones(N, device="cuda", requires_grad=True), torch.zeros(N, device="cuda", requires_grad=True)
    rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    y = _Dw3x3Fn.apply(x, w, None)
    dy = torch.randn_like(y)
    z = _Pointwise1x1Fn.apply(y, pw, None)
    dz = torch.randn_like(z)
    r = _BnReluFn.apply(z, g, b, rm, rv, 0.1, 1e-5, True)
    print(f"B{B} {H}^2 C{C}->{N}: dw fwd {t(lambda: _Dw3x3Fn.apply(x, w, None)):7.1f} us | dw bwd (dx+dw) "
          f"{t(lambda: torch.autograd.grad(y, [x, w], dy, retain_graph=True)):7.1f} | pw fwd {t(lambda: _Pointwise1x1Fn.apply(y, pw, None)):7.1f}"
          f" | pw bwd {t(lambda: torch.autograd.grad(z, [y, pw], dz, retain_graph=True)):7.1f} | bn fwd "
          f"{t(lambda: _BnReluFn.apply(z, g, b, rm, rv, 0.1, 1e-5, True)):7.1f} | bn bwd "
          f"{t(lambda: torch.autograd.grad(r, [z, g, b], dz, retain_graph=True)):7.1f}")
    xn = x.detach().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    conv = torch.nn.Conv2d(C, C, 3, padding=1, groups=C, bias=False).cuda()
    yn = conv(xn)
    print(f"      torch NCHW depthwise fwd {t(lambda: conv(xn)):7.1f} us, bwd {t(lambda: torch.autograd.grad(yn, [xn, conv.weight], torch.ones_like(yn), retain_graph=True)):7.1f}")
