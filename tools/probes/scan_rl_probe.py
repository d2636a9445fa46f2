"""Probe (GPU box): forward time of the channels-on-lanes cross-scan with parts switched off (nnz_scan_tuning knob 2:
bit 0 no y store, bit 1 no checkpoint store, bit 2 no u reload) - which memory stream costs what."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnuzoo_amd._lib import call, load, ptr, stream_ptr
lib = load()
Di, H = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 512
B, W, R, N = 2, H, max(1, Di // 32), 16
L, Cp, K = H * W, R + 32, 4
f = dict(device="cuda", dtype=torch.float32)
x2 = torch.randn(2, B, Di, L, **f); P = torch.randn(2, B, 2 * Cp, L, **f) * 0.5
Wdt = torch.randn(K * Di, R, **f) * 0.3; Alog = torch.randn(K * Di, N, **f) * 0.3
Dv, bias = torch.randn(K * Di, **f), torch.randn(K * Di, **f)
y = torch.empty(B, K * Di, L, **f)
state = torch.empty(lib.nnz_ss2d_scan_state_floats(B, Di, L), **f)
ws = torch.empty(lib.nnz_ss2d_scan_workspace_floats(B, Di, L), **f)
fwd = lambda: call("nnz_ss2d_scan_forward", ptr(x2), ptr(P), ptr(Wdt), ptr(Alog), ptr(Dv), ptr(bias), ptr(y), ptr(state), ptr(ws), B, Di, R, L, 1, 1, stream_ptr())
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for clb in (16, 8, 4):
    call("nnz_scan_tuning", 1, clb)
    for dbg in (0, 1, 2, 3, 4, 7):
        call("nnz_scan_tuning", 2, dbg)
        print(f"clb {clb} dbg {dbg}: fwd {t(fwd):.3f} ms", flush=True)
