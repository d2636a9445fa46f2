"""Probe (GPU box): what the streaming kernels of the 3-D step could reach - torch's device copy (1 read + 1 write) against
norm_kernel<1> (apply, 1 read + 1 write), <2> (backward reduce, 2 reads) and <3> (backward apply, 2 reads + 1 write) on the
full-resolution activation (2 x 128^3 x 32 ch fp16 = 268 MB)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnuzoo_amd import hip_ops as ops

def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

for N, V, C in [(2, 128 ** 3, 32), (2, 64 ** 3, 64), (2, 32 ** 3, 128)]:
    x = torch.randn(N, V, C, device="cuda").half()
    g = torch.randn(N, V, C, device="cuda").half()
    y = torch.empty_like(x)
    nb = x.numel() * 2
    tc = t(lambda: y.copy_(x))
    print(f"N={N} V={V} C={C}: tensor {nb/1e6:.0f} MB; torch copy {tc*1e3:.0f} us = {2*nb/tc/1e9:.2f} TB/s", flush=True)
    stats = torch.zeros(N, C, 2, device="cuda"); red = torch.zeros(N, C, 2, device="cuda")
    gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
    dx = torch.empty_like(x)
    ops.instnorm_stats(x, stats, N, V, C, C)
    from nnuzoo_amd._lib import call, ptr, stream_ptr
    for blocks in (1024, 2048, 4096, 8192, 16384):
        call("nnz_norm_tuning", 0, blocks)
        ta = t(lambda: ops.instnorm_lrelu_apply(x, stats, gamma, beta, y, N, V, C, C, C, 1e-5, 0.01))
        tr = t(lambda: call("nnz_instnorm_lrelu_bwd_reduce", ptr(x), ptr(g), ptr(stats), ptr(gamma), ptr(beta), ptr(red), N, V, C, C, C, 1e-5, 0.01, 0, stream_ptr()))
        tb = t(lambda: call("nnz_instnorm_lrelu_bwd_apply", ptr(x), ptr(g), ptr(stats), ptr(red), ptr(gamma), ptr(beta), ptr(dx), N, V, C, C, C, C, 1e-5, 0.01, None, None, stream_ptr()))
        print(f"   blocks {blocks:6d}: apply {ta*1e3:5.0f} us {2*nb/ta/1e9:.2f} TB/s | bwd reduce {tr*1e3:5.0f} us {2*nb/tr/1e9:.2f} | bwd apply {tb*1e3:5.0f} us {3*nb/tb/1e9:.2f}", flush=True)
    call("nnz_norm_tuning", 0, 2048)
