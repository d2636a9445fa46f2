"""Per-layer timing of the tap-table conv kernels on the 3d_fullres 128^3 PlainConvUNet shapes (batch 2).
Usage (GPU box): python tools/bench_conv_layers.py [--reps 10]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnuzoo_amd import conv_plan as cp
from nnuzoo_amd import hip_ops as ops
from nnuzoo_amd.hip_ops import PreparedTable

LAYERS = [  # name, cin, cout, in edge, stride
    ("enc0.1", 32, 32, 128, 1), ("dec0.0", 64, 32, 128, 1), ("enc1.0", 32, 64, 128, 2), ("enc1.1", 64, 64, 64, 1),
    ("dec1.0", 128, 64, 64, 1), ("enc2.0", 64, 128, 64, 2), ("enc2.1", 128, 128, 32, 1), ("dec2.0", 256, 128, 32, 1),
    ("enc3.0", 128, 256, 32, 2), ("enc3.1", 256, 256, 16, 1), ("dec3.0", 512, 256, 16, 1), ("enc4.0", 256, 320, 16, 2),
    ("enc4.1", 320, 320, 8, 1), ("dec4.0", 640, 320, 8, 1), ("enc5.0", 320, 320, 8, 2), ("enc5.1", 320, 320, 4, 1),
]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--tuning", type=str, default="", help="comma list knob=value for nnz_conv_tuning (include/nnuzoo_hip.h)")
    ap.add_argument("--innorm", type=int, default=0, help="1: x is a raw conv output normalised by the consumers (round 4): "
                    "forward with an input table, weight gradient in the flipped form with a plain-operand table")
    a = ap.parse_args()
    if a.tuning:
        from nnuzoo_amd import _lib
        for kv in a.tuning.split(","):
            k, v = kv.split("=")
            _lib.call("nnz_conv_tuning", int(k), int(v))
        print("tuning:", a.tuning)
    N = 2
    dev = "cuda"
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    totf = 0.0
    for name, cin, cout, edge, stride in LAYERS:
        if a.only and a.only not in name:
            continue
        dims = (edge,) * 3
        od = cp.conv_out_dims(dims, (3, 3, 3), stride)
        V, Vo = edge ** 3, int(np.prod(od))
        flops = 2.0 * N * Vo * cin * cout * 27
        x = torch.randn(N, V, cin, device=dev).to(torch.float16)
        dy = torch.randn(N, Vo, cout, device=dev).to(torch.float16)
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        y = torch.empty(N, Vo, cout, device=dev, dtype=torch.float16)
        dx = torch.empty(N, V, cin, device=dev, dtype=torch.float16)
        dw = torch.empty(27, cin, cout, device=dev, dtype=torch.float32)
        pf = PreparedTable(cp.conv_forward(N, dims, cin, cout, stride=stride))
        pd = PreparedTable(cp.conv_dgrad(N, dims, cin, cout, stride=stride))
        pw = PreparedTable(cp.conv_wgrad(N, dims, cin, cout, stride=stride))
        wf = ops.pack_weight(w, pf, cin, cout, 27, cin * 27, 1)
        wd = ops.pack_weight(w, pd, cout, cin, cin * 27, 27, 1)
        inn = None
        if a.innorm:
            tab = torch.randn(N, cin, 4, device=dev)
            tab[:, :, 2] = 1.0 + 0.1 * tab[:, :, 2]
            inn = ops.InNorm(tab, 0.01)
        tf = timeit(lambda: ops.conv_tap_forward(pf, x, wf, None, y, innorm=inn), a.reps)
        td = timeit(lambda: ops.conv_tap_forward(pd, dy, wd, None, dx), a.reps)
        # the schedule's weight-gradient path: partial blocks + fixed-order reduction into the torch-layout gradient
        ws = torch.empty(ops.conv_tap_wgrad_workspace_floats(pw), device=dev, dtype=torch.float32)
        gw = torch.empty_like(w)
        if a.innorm and stride == 1:
            pwf = PreparedTable(cp.conv_wgrad_flipped(N, dims, cin, cout))
            tw = timeit(lambda: ops.conv_tap_wgrad_to_grad(pwf, dy, x, ws, gw, cin * 27, 27, 1, plain_norm=inn), a.reps)
        elif a.innorm:
            tw = timeit(lambda: ops.conv_tap_wgrad_to_grad(pw, x, dy, ws, gw, 27, cin * 27, 1, boxed_norm=inn), a.reps)
        else:
            tw = timeit(lambda: ops.conv_tap_wgrad_to_grad(pw, x, dy, ws, gw, 27, cin * 27, 1), a.reps)
        tot["fwd"] += tf; tot["dgrad"] += td; tot["wgrad"] += tw
        totf += flops
        print(f"{name:8s} {cin:4d}->{cout:4d} @{edge:3d} s{stride}  {flops/1e9:8.1f} GF | fwd {tf*1e3:8.3f} ms {flops/tf/1e12:7.1f} TF/s"
              f" | dgrad {td*1e3:8.3f} ms {flops/td/1e12:7.1f} | wgrad {tw*1e3:8.3f} ms {flops/tw/1e12:7.1f}", flush=True)
    print(f"TOTAL {totf/1e9:.1f} GF/pass: fwd {tot['fwd']*1e3:.2f} ms, dgrad {tot['dgrad']*1e3:.2f} ms, wgrad {tot['wgrad']*1e3:.2f} ms"
          f" -> {3*totf/sum(tot.values())/1e12:.1f} TF/s overall")


if __name__ == "__main__":
    main()
