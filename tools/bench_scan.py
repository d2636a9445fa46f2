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


for B, KD, L in ([] if "--xs-only" in sys.argv else SHAPES):
    K, N = 4, 16
    u = torch.randn(B, KD, L, device="cuda", requires_grad=True)
    dl = (torch.randn(B, KD, L, device="cuda") * 0.5).requires_grad_(True)
    A = (-torch.exp(torch.randn(KD, N, device="cuda") * 0.3)).requires_grad_(True)
    Bm = torch.randn(B, K, N, L, device="cuda", requires_grad=True)
    Cm = torch.randn(B, K, N, L, device="cuda", requires_grad=True)
    D = torch.randn(KD, device="cuda", requires_grad=True)
    bias = torch.randn(KD, device="cuda", requires_grad=True)
    dy = torch.randn(B, KD, L, device="cuda")
    from nnuzoo_amd._lib import call as _call, load as _load
    fb = 4 * (3 * B * KD * L + 2 * B * K * N * L)
    bb = 4 * (5 * B * KD * L + 4 * B * K * N * L)
    for gen in (0, 1):     # 0: time-on-lanes kernels only; 1: channels-on-lanes kernels where the launcher picks them
        _call("nnz_scan_tuning", 0, gen)
        n0 = _load().nnz_scan_tuning_get(3)
        with torch.no_grad():
            tf = t(lambda: selective_scan_fn(u, dl, A, Bm, Cm, D, None, bias, True))
        y = selective_scan_fn(u, dl, A, Bm, Cm, D, None, bias, True)
        tb = t(lambda: torch.autograd.grad(y, [u, dl, A, Bm, Cm, D, bias], dy, retain_graph=True))
        took = "channels-on-lanes" if _load().nnz_scan_tuning_get(3) > n0 else "time-on-lanes"
        print(f"({B},{KD},{L}) gen{gen} [{took}]: fwd {tf:8.3f} ms {fb/tf/1e6:8.1f} GB/s | bwd {tb:8.3f} ms "
              f"{bb/tb/1e6:8.1f} GB/s", flush=True)
        if took == "time-on-lanes" and gen == 1:
            break
    _call("nnz_scan_tuning", 0, 1)

# ---- cross-scan mode (the fused SS2D core): the C-ABI scan alone on M2Net's (B, Di, H, W) shapes -------------------------
# algorithmic bytes: fwd reads x2 once per source (2 B D L) and P (2 B 2Cp L), writes y (B 4D L); bwd reads x2, P, dy2,
# writes du (B 4D L) and dP.  state-updates = B * 4 D * 16 * L per pass.
from nnuzoo_amd._lib import call, load, ptr, stream_ptr

print("# cross-scan mode (nnz_ss2d_scan_forward / _backward), B=2: ms, algorithmic GB/s, G state-updates/s")
lib = load()
for Di, H in [(32, 512), (64, 256), (32, 256), (128, 128), (256, 64), (256, 32), (32, 16)]:
    B, W, R, N = 2, H, max(1, Di // 32), 16
    L, Cp, K = H * W, R + 32, 4
    f = dict(device="cuda", dtype=torch.float32)
    x2 = torch.randn(2, B, Di, L, **f)
    P = torch.randn(2, B, 2 * Cp, L, **f) * 0.5
    Wdt = torch.randn(K * Di, R, **f) * 0.3
    Alog = torch.randn(K * Di, N, **f) * 0.3
    Dv, bias = torch.randn(K * Di, **f), torch.randn(K * Di, **f)
    y, du = torch.empty(B, K * Di, L, **f), torch.empty(B, K * Di, L, **f)
    dy2 = torch.randn(2, B, Di, L, **f)
    dP = torch.empty_like(P)
    dWdt, dA, dD, dbias = torch.empty_like(Wdt), torch.empty_like(Alog), torch.empty_like(Dv), torch.empty_like(bias)
    state = torch.empty(lib.nnz_ss2d_scan_state_floats(B, Di, L), **f)
    gstate = torch.empty(lib.nnz_ss2d_scan_grad_state_floats(B, Di, L), **f)
    ws = torch.empty(lib.nnz_ss2d_scan_workspace_floats(B, Di, L), **f)
    fwd = lambda: call("nnz_ss2d_scan_forward", ptr(x2), ptr(P), ptr(Wdt), ptr(Alog), ptr(Dv), ptr(bias), ptr(y),
                       ptr(state), ptr(ws), B, Di, R, L, 1, 1, stream_ptr())
    bwd = lambda: call("nnz_ss2d_scan_backward", ptr(x2), ptr(P), ptr(Wdt), ptr(Alog), ptr(Dv), ptr(bias), ptr(dy2),
                       ptr(state), ptr(gstate), ptr(ws), ptr(du), ptr(dP), ptr(dWdt), ptr(dA), ptr(dD), ptr(dbias), B,
                       Di, R, L, 1, 1, stream_ptr())
    fb = 4 * B * L * (2 * Di + 2 * 2 * Cp + 4 * Di)
    bb = 4 * B * L * (2 * Di + 2 * 2 * Cp + 2 * Di + 4 * Di + 2 * 2 * Cp)
    su = B * K * Di * N * L
    outs = {}
    for gen in (0, 1):          # 0: time-on-lanes kernels (selective_scan.hip), 1: channels-on-lanes (ss2d_scan_rl.hpp)
        call("nnz_scan_tuning", 0, gen)
        tf = t(fwd)
        tb = t(bwd)
        torch.cuda.synchronize()
        outs[gen] = [v.clone() for v in (y, du, dP, dWdt, dA, dD, dbias)]
        print(f"Di={Di:4d} {H}x{W} gen{gen}: fwd {tf:7.3f} ms {fb/tf/1e6:7.1f} GB/s {su/tf/1e6:7.1f} Gsu/s | "
              f"bwd {tb:7.3f} ms {bb/tb/1e6:7.1f} GB/s {su/tb/1e6:7.1f} Gsu/s", flush=True)
    rel = [((p - q).abs().max() / q.abs().max().clamp_min(1e-20)).item() for p, q in zip(outs[1], outs[0])]
    print("      gen1 vs gen0 max|diff|/max|ref|: " + " ".join(f"{n}={v:.1e}" for n, v in
                                                                 zip(("y", "du", "dP", "dWdt", "dA", "dD", "dbias"), rel)), flush=True)
