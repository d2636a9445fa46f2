"""Timing of the HIP selective scan on the M2Net@512^2 call shapes (B=2).  Reports ms and algorithmic GB/s
(fwd bytes = 4*(3*B*KD*L + 2*B*K*N*L); bwd = reads u, delta, B, C, dy + writes du, ddelta, dB, dC)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnuzoo_amd.selective_scan import selective_scan_fn

SHAPES = [(2, 128, 262144), (2, 256, 65536), (2, 128, 65536), (2, 512, 16384), (2, 1024, 4096), (2, 1024, 1024),
          (2, 512, 256), (2, 128, 256)]


def t(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for B, KD, L in SHAPES:
    K, N = 4, 16
    u = torch.randn(B, KD, L, device="cuda", requires_grad=True)
    dl = (torch.randn(B, KD, L, device="cuda") * 0.5).requires_grad_(True)
    A = (-torch.exp(torch.randn(KD, N, device="cuda") * 0.3)).requires_grad_(True)
    Bm = torch.randn(B, K, N, L, device="cuda", requires_grad=True)
    Cm = torch.randn(B, K, N, L, device="cuda", requires_grad=True)
    D = torch.randn(KD, device="cuda", requires_grad=True)
    bias = torch.randn(KD, device="cuda", requires_grad=True)
    dy = torch.randn(B, KD, L, device="cuda")
    with torch.no_grad():
        tf = t(lambda: selective_scan_fn(u, dl, A, Bm, Cm, D, None, bias, True))
    y = selective_scan_fn(u, dl, A, Bm, Cm, D, None, bias, True)
    tb = t(lambda: torch.autograd.grad(y, [u, dl, A, Bm, Cm, D, bias], dy, retain_graph=True))
    fb = 4 * (3 * B * KD * L + 2 * B * K * N * L)
    bb = 4 * (5 * B * KD * L + 4 * B * K * N * L)
    print(f"({B},{KD},{L}): fwd {tf:8.3f} ms {fb/tf/1e6:8.1f} GB/s | bwd {tb:8.3f} ms {bb/tb/1e6:8.1f} GB/s", flush=True)
