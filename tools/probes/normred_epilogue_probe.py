"""Where does the fused norm-backward epilogue of the data-gradient launches spend its time?  Plain launch vs
nnz_conv_tap_dgrad_normred with parts of the epilogue switched off (nnz_conv_tuning knob 6: 1 no x loads, 2 no sums,
4 no fixed-point adds / ticket).  Results with bits set are WRONG on purpose.
Usage (GPU box): python tools/probes/normred_epilogue_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnuzoo_amd import _lib
from nnuzoo_amd import conv_plan as cp
from nnuzoo_amd import hip_ops as ops
from nnuzoo_amd.hip_ops import PreparedTable

CASES = [("enc0.1 dgrad", 32, 32, 128, 1, False), ("dec0.0 dgrad", 64, 32, 128, 1, False),
         ("enc1.1 dgrad", 64, 64, 64, 1, False), ("enc1.0 dgrad s2 acc", 32, 64, 128, 2, True)]


def timeit(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    N, dev = 2, "cuda"
    for name, cin, cout, edge, stride, acc in CASES:
        dims = (edge,) * 3
        ldo = 2 * cin if acc else cin
        pd = PreparedTable(cp.conv_dgrad(N, dims, cin, cout, stride=stride, ldi=cout, ldo=ldo, accumulate=acc))
        od = cp.conv_out_dims(dims, (3, 3, 3), stride)
        V, Vo = edge ** 3, int(np.prod(od))
        dy = torch.randn(N, Vo, cout, device=dev).to(torch.float16)
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        wd = ops.pack_weight(w, pd, cout, cin, cin * 27, 27, 1)
        buf = torch.zeros(N, V, ldo, device=dev, dtype=torch.float16)
        dx = buf[:, :, ldo - cin:]
        x_raw = torch.randn(N, V, cin, device=dev).to(torch.float16)
        sc = ops.NormScratch(torch.device(dev), N * max(cin, cout))
        nstat = torch.empty((N, cin, 4), device=dev)
        g, b = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
        ops.instnorm_stats_det(x_raw, N, V, cin, cin, sc, g, b, 1e-5, nstat=nstat)
        nred = torch.empty((N, cin, 2), device=dev)
        dg, db = torch.empty(cin, device=dev), torch.empty(cin, device=dev)
        dxo = torch.empty(N, V, cin, device=dev, dtype=torch.float16)
        t0 = timeit(lambda: ops.conv_tap_forward(pd, dy, wd, None, dx))
        tr = timeit(lambda: ops.instnorm_lrelu_bwd_tab(x_raw, dx, nstat, sc, nred, dxo, N, V, cin, cin, ldo, cin, 0.01,
                                                       dgamma=dg, dbeta=db)) - \
            timeit(lambda: ops.instnorm_lrelu_bwd_apply_tab(x_raw, dx, nstat, nred, dxo, N, V, cin, cin, ldo, cin, 0.01))
        line = f"{name:20s} plain {t0:7.1f} us | separate reduce {tr:6.1f} us | fused:"
        for bits in (0, 1, 2, 3, 4, 7):
            _lib.call("nnz_conv_tuning", 6, bits)
            t = timeit(lambda: ops.conv_tap_dgrad_normred(pd, dy, wd, dx, x_raw, cin, nstat, 0.01, sc, nred, dg, db))
            if bits & 4:      # the accumulators were not emptied by a last workgroup
                sc.acc.zero_(); sc.counter.zero_()
            line += f"  [{bits}] {t:7.1f}"
        _lib.call("nnz_conv_tuning", 6, 0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
