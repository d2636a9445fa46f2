import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnuzoo_amd._lib import call, load, ptr, stream_ptr
lib = load()
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
call("nnz_scan_tuning", 2, 0)
for Di, H in [(256, 64), (128, 64), (256, 32), (64, 64), (128, 32), (256, 16), (32, 64), (64, 32), (128, 16), (32, 32), (64, 16), (32, 16), (256, 8)]:
    B, W, R, N = 2, H, max(1, Di // 32), 16
    L, Cp, K = H * W, R + 32, 4
    f = dict(device="cuda", dtype=torch.float32)
    x2 = torch.randn(2, B, Di, L, **f); P = torch.randn(2, B, 2 * Cp, L, **f) * 0.5
    Wdt = torch.randn(K * Di, R, **f) * 0.3; Alog = torch.randn(K * Di, N, **f) * 0.3
    Dv, bias = torch.randn(K * Di, **f), torch.randn(K * Di, **f)
    y, du = torch.empty(B, K * Di, L, **f), torch.empty(B, K * Di, L, **f)
    dy2 = torch.randn(2, B, Di, L, **f); dP = torch.empty_like(P)
    dWdt, dA, dD, dbias = torch.empty_like(Wdt), torch.empty_like(Alog), torch.empty_like(Dv), torch.empty_like(bias)
    state = torch.empty(lib.nnz_ss2d_scan_state_floats(B, Di, L), **f)
    gstate = torch.empty(lib.nnz_ss2d_scan_grad_state_floats(B, Di, L), **f)
    ws = torch.empty(lib.nnz_ss2d_scan_workspace_floats(B, Di, L), **f)
    fwd = lambda: call("nnz_ss2d_scan_forward", ptr(x2), ptr(P), ptr(Wdt), ptr(Alog), ptr(Dv), ptr(bias), ptr(y), ptr(state), ptr(ws), B, Di, R, L, 1, 1, stream_ptr())
    bwd = lambda: call("nnz_ss2d_scan_backward", ptr(x2), ptr(P), ptr(Wdt), ptr(Alog), ptr(Dv), ptr(bias), ptr(dy2), ptr(state), ptr(gstate), ptr(ws), ptr(du), ptr(dP), ptr(dWdt), ptr(dA), ptr(dD), ptr(dbias), B, Di, R, L, 1, 1, stream_ptr())
    r = {}
    for gen in (0, 1):
        call("nnz_scan_tuning", 0, gen)
        r[gen] = (t(fwd), t(bwd))
    print(f"Di={Di:4d} {H:3d}^2 rowsteps {B*K*Di*L/2**20:6.2f}M  gen0 fwd {r[0][0]*1e3:6.1f} bwd {r[0][1]*1e3:6.1f} us | gen1 fwd {r[1][0]*1e3:6.1f} bwd {r[1][1]*1e3:6.1f} | sum {1e3*sum(r[0]):6.1f} -> {1e3*sum(r[1]):6.1f}", flush=True)
