"""Diagnostic (GPU box): hipGraph replay of (a) the LN backward kernel alone (no memset nodes), (b) memset nodes alone,
(c) both, with small eager allocations filled with NaN between replays."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nnuzoo_amd._lib import call, ptr, stream_ptr

torch.manual_seed(0)
R, C = 1152, 32
x = torch.randn(R, C, device="cuda")
dy = torch.randn(R, C, device="cuda")
w = torch.ones(C, device="cuda")
mean, rstd = x.mean(1).contiguous(), (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
dx, dw, db = torch.empty_like(x), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
z = torch.empty(4096, device="cuda")


def kernel_only():
    call("nnz_layer_norm_backward", ptr(x), 0, ptr(w), ptr(mean), ptr(rstd), ptr(dy), 0, ptr(dx), 0, 0, 0, R, C, stream_ptr())


def memset_only():
    z.zero_()


def both():
    call("nnz_layer_norm_backward", ptr(x), 0, ptr(w), ptr(mean), ptr(rstd), ptr(dy), 0, ptr(dx), ptr(dw), ptr(db), 0, R, C,
         stream_ptr())


def junk():
    j = [torch.full((1 + 37 * i,), float("nan"), device="cuda") for i in range(3000)]
    torch.cuda.synchronize()
    del j


for name, fn, outs in (("kernel_only", kernel_only, [dx]), ("memset_only", memset_only, [z]), ("both", both, [dx, db])):
    fn()
    torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fn()
    for rep in range(4):
        for o in outs:
            o.fill_(777.0)
        g.replay()
        torch.cuda.synchronize()
        errs = [(o - r).abs().max().item() for o, r in zip(outs, ref)]
        print(name, "replay", rep, "max err", errs, flush=True)
        junk()
