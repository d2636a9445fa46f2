"""Window-attention core (csrc/window_attention.hip) on the SwT2Net stage shapes at 512^2, batch 2: per-launch time, algorithmic
FLOP rate (4 * 49^2 * head_dim per (window, head) forward, 2.5x that backward) and algorithmic bytes (qkv + out forward; qkv +
dout + dqkv backward) against the fp32 MFMA peak (157 TFLOP/s) and HBM (8 TB/s).
Usage (GPU box): python tools/bench_window_attention.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnuzoo_amd._lib import call, ptr, stream_ptr
from nnuzoo_amd.hip_ops import det_scratch


def timeit(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def main():
    ar = torch.arange(7)
    yy, xx = torch.meshgrid(ar, ar, indexing="ij")
    y, x = yy.flatten(), xx.flatten()
    idx = ((y[:, None] - y[None, :] + 6) * 13 + (x[:, None] - x[None, :] + 6)).to(torch.int32).cuda()
    for H, heads in [(133, 3), (70, 6), (35, 12), (21, 24), (14, 24), (7, 24)]:
        B, hd = 2, 32
        C = heads * hd
        nwin = B * (H // 7) ** 2
        qkv = torch.randn(B, H, H, 3 * C, device="cuda")
        table = torch.randn(169, heads, device="cuda") * 0.5
        dout = torch.randn(B, H, H, C, device="cuda")
        out, dqkv, dtable = torch.empty_like(dout), torch.empty_like(qkv), torch.empty_like(table)
        sc = det_scratch(qkv.device, 170 * heads)
        for shift in (0, 3):
            # the C-ABI entry points directly, 20 launches back to back (through autograd the small shapes are host-bound)
            tf = timeit(lambda: call("nnz_window_attention_forward", ptr(qkv), ptr(table), ptr(idx), ptr(out), B, H, H, C,
                                     heads, shift, hd ** -0.5, stream_ptr()))
            tb = timeit(lambda: call("nnz_window_attention_backward", ptr(qkv), ptr(table), ptr(idx), ptr(dout), ptr(dqkv),
                                     ptr(dtable), ptr(sc.acc), ptr(sc.counter), B, H, H, C, heads, shift, hd ** -0.5,
                                     stream_ptr()))
            fl = 4.0 * 49 * 49 * hd * nwin * heads
            by_f = 4.0 * B * H * H * (3 * C + C)
            by_b = 4.0 * B * H * H * (3 * C + C + 3 * C)
            print(f"{H:4d}^2 x {heads:2d} heads shift {shift}: {nwin * heads:5d} (window, head) | fwd {tf * 1e6:7.1f} us "
                  f"{fl / tf / 1e12:6.2f} TFLOP/s {by_f / tf / 1e9:7.0f} GB/s | bwd {tb * 1e6:7.1f} us "
                  f"{2.5 * fl / tb / 1e12:6.2f} TFLOP/s {by_b / tb / 1e9:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
