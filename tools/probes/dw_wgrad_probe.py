"""A/B of the depthwise 3x3 weight gradient: csrc/depthwise_wgrad.hip vs ATen's direct kernel (NNZ_DW_WGRAD=0), on the
SSND2Net / LightMamba2Net layer shapes."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnuzoo_amd.nets.common2d import _DepthwiseNativeFn


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for dt in (torch.float16, torch.float32):
    for B, C, H in [(2, 32, 512), (2, 64, 256), (2, 128, 128), (2, 256, 64), (2, 512, 32), (2, 512, 16), (2, 16, 512)]:
        x = torch.randn(B, C, H, H, device="cuda", dtype=dt)
        w = torch.randn(C, 1, 3, 3, device="cuda", dtype=dt, requires_grad=True)
        dy = torch.randn(B, C, H, H, device="cuda", dtype=dt)
        res = {}
        for mode in ("0", "1"):
            os.environ["NNZ_DW_WGRAD"] = mode
            y = _DepthwiseNativeFn.apply(x, w, None, (1, 1), (1, 1), (1, 1), C)
            res[mode] = t(lambda: torch.autograd.grad(y, [w], dy, retain_graph=True))
        gb = 2 * x.numel() * x.element_size() / 1e3
        print(f"{str(dt)[6:]:8s} B{B} C{C:4d} {H}x{H}: ATen {res['0']:8.1f} us | hip {res['1']:8.1f} us  ({gb / res['1']:7.1f} GB/s)", flush=True)
