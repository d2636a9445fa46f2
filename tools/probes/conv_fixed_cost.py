"""Diagnostic (GPU box): per-launch fixed cost of conv_box from the forward time at fixed Cout / tile and growing Cin
(time = fixed + slices * per_slice).  Usage: python tools/probes/conv_fixed_cost.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnuzoo_amd import conv_plan as cp
from nnuzoo_amd import hip_ops as ops
from nnuzoo_amd.hip_ops import PreparedTable


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


N = 2
for cout, edge in [(32, 128), (64, 64)]:
    ts = []
    for cin in (32, 64, 128, 256):
        dims = (edge,) * 3
        V = edge ** 3
        x = torch.randn(N, V, cin, device="cuda").to(torch.float16)
        w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
        y = torch.empty(N, V, cout, device="cuda", dtype=torch.float16)
        pf = PreparedTable(cp.conv_forward(N, dims, cin, cout, stride=1))
        wf = ops.pack_weight(w, pf, cin, cout, 27, cin * 27, 1)
        t = timeit(lambda: ops.conv_tap_forward(pf, x, wf, None, y))
        ts.append((cin // 16, t))
        print(f"cout {cout} @{edge}^3 cin {cin:4d} ({cin // 16} slices): {t * 1e6:8.1f} us  {2.0 * N * V * cin * cout * 27 / t / 1e12:7.1f} TF/s",
              flush=True)
    k = np.array([a for a, _ in ts], dtype=float)
    tt = np.array([b for _, b in ts]) * 1e6
    slope, icpt = np.polyfit(k, tt, 1)
    print(f"  -> fixed {icpt:.1f} us + {slope:.1f} us per 16-channel slice (fixed = {icpt / slope:.2f} slices' worth)", flush=True)
