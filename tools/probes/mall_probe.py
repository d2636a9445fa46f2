"""probe: does a tensor that was just written (134 MB: one sample of a 32-channel 128^3 activation) get read faster than a
cold one?  (256 MiB Infinity Cache; decides whether per-sample scheduling of the full-resolution layers would pay)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nnuzoo_amd import hip_ops as ops

dev = torch.device("cuda")
C = 32
for N, V in [(1, 128 ** 3), (2, 128 ** 3), (1, 64 ** 3 * 2), (2, 64 ** 3)]:
    x = torch.randn(N, V, C, device=dev).half()
    y = torch.empty_like(x)
    nstat = torch.rand(N, C, 4, device=dev)
    junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev)

    def t(fn, prep):
        ts = []
        for _ in range(5):
            prep()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        return sorted(ts)[2]

    apply = lambda: ops.instnorm_lrelu_apply_tab(x, nstat, y, N, V, C, C, C, 0.01)
    cold = t(apply, lambda: junk.fill_(1))
    warm = t(apply, lambda: x.copy_(x.clone()) if False else x.mul_(1.0))     # x just written by the previous kernel
    both = t(apply, lambda: (x.mul_(1.0), y.mul_(1.0)))
    mb = x.numel() * 2 / 1e6
    print(f"N={N} V={V} ({mb:.0f} MB in, {mb:.0f} MB out): apply cold {cold:.1f} us ({2*mb/cold*1e-3:.2f} TB/s) | input just written {warm:.1f} us "
          f"({2*mb/warm*1e-3:.2f} TB/s) | input and output just written {both:.1f} us ({2*mb/both*1e-3:.2f} TB/s)")
