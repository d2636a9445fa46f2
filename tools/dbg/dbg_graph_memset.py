"""Diagnostic (GPU box): are hipMemsetAsync nodes inside our C functions replayed (and ordered) under hipGraph?
No autograd here: the C-ABI LayerNorm backward (memset of dgamma/dbeta + atomics kernel) captured directly."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nnuzoo_amd._lib import call, ptr, stream_ptr

junk_on = len(sys.argv) > 1 and sys.argv[1] == "junk"
torch.manual_seed(0)
R, C = 1152, 32
x = torch.randn(R, C, device="cuda")
dy = torch.randn(R, C, device="cuda")
w = torch.ones(C, device="cuda")
mean, rstd = x.mean(1).contiguous(), (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
dx, dw, db = torch.empty_like(x), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")


def run():
    call("nnz_layer_norm_backward", ptr(x), 0, ptr(w), ptr(mean), ptr(rstd), ptr(dy), 0, ptr(dx), ptr(dw), ptr(db), 0, R, C,
         stream_ptr())


run()
torch.cuda.synchronize()
print("eager db[0:3]", db[:3].tolist(), "ref", dy.sum(0)[:3].tolist(), flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    run()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    run()
for rep in range(5):
    db.fill_(777.0)
    g.replay()
    torch.cuda.synchronize()
    print("replay", rep, "db[0:3]", db[:3].tolist(), flush=True)
    if junk_on:
        j = [torch.full((1 + 37 * i,), float("nan"), device="cuda") for i in range(3000)]
        torch.cuda.synchronize()
        del j
