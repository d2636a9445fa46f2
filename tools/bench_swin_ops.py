"""Per-launch times of the kernels of one Swin block (fp32, csrc/dense32.hip, layer_norm.hip, window_attention.hip, residual.hip) on
the twelve (tokens, channels, heads) levels of SwT2Net at 512^2, batch 2 - the census DESIGN.md section 4 prices the block fusion
with.  Every op through its C-ABI entry point, 20 launches back to back (through autograd the small shapes are host-bound).
Columns: microseconds per launch; `blocks` = how many Swin blocks of the net run at that level.
Usage (GPU box): python tools/bench_swin_ops.py [--fused]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnuzoo_amd._lib import call, ptr, stream_ptr
from nnuzoo_amd.hip_ops import det_scratch

# (unpadded token grid edge, channels, heads, number of Swin blocks at this level in SwT2Net)
LEVELS = [(128, 32, 2, 8), (64, 64, 2, 8), (64, 64, 4, 8), (64, 96, 3, 16), (32, 128, 4, 16), (32, 128, 8, 8), (32, 192, 6, 16),
          (16, 256, 8, 4), (16, 256, 16, 16), (16, 384, 12, 32), (8, 512, 16, 4), (8, 768, 24, 8)]


def timeit(fn, reps=20):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    B = a.batch
    ar = torch.arange(7)
    yy, xx = torch.meshgrid(ar, ar, indexing="ij")
    y, x = yy.flatten(), xx.flatten()
    idx = ((y[:, None] - y[None, :] + 6) * 13 + (x[:, None] - x[None, :] + 6)).to(torch.int32).cuda()
    names = ["pad", "ln_f", "qkv", "attn_f", "proj", "res_f", "fc1g", "fc2", "crop", "fc2_dg", "fc1_dg", "ln_b", "proj_dg",
             "attn_b", "qkv_dg", "res_b"]
    print(f"{'level':>22s} {'blk':>3s} " + " ".join(f"{n:>7s}" for n in names) + "   fwd_sum  bwd_sum")
    tot_f = tot_b = 0.0
    for H, C, heads, blocks in LEVELS:
        py = 7 - H % 7
        Hp = H + py
        T, Tp = B * H * H, B * Hp * Hp
        dev = "cuda"
        xs = torch.randn(B, H, H, C, device=dev)
        xp = torch.randn(B, Hp, Hp, C, device=dev)
        g, bta = torch.randn(C, device=dev), torch.randn(C, device=dev)
        yln, mean, rstd = torch.empty_like(xp), torch.empty(Tp, device=dev), torch.empty(Tp, device=dev)
        wqkv, bqkv = torch.randn(3 * C, C, device=dev) * 0.05, torch.randn(3 * C, device=dev)
        qkv = torch.empty(Tp, 3 * C, device=dev)
        table = torch.randn(169, heads, device=dev) * 0.5
        ao = torch.empty(Tp, C, device=dev)
        wp, bp = torch.randn(C, C, device=dev) * 0.05, torch.randn(C, device=dev)
        po = torch.empty(Tp, C, device=dev)
        w1, b1 = torch.randn(4 * C, C, device=dev) * 0.05, torch.randn(4 * C, device=dev)
        w2, b2 = torch.randn(C, 4 * C, device=dev) * 0.05, torch.randn(C, device=dev)
        h, act = torch.empty(Tp, 4 * C, device=dev), torch.empty(Tp, 4 * C, device=dev)
        dqkv, dtab = torch.empty_like(qkv), torch.empty_like(table)
        dx, dw, db = torch.empty_like(xp), torch.empty(C, device=dev), torch.empty(C, device=dev)
        sc = det_scratch(xp.device, max(170 * heads, 2 * C))
        r = {}
        r["pad"] = timeit(lambda: call("nnz_pad_top_left", ptr(xs), ptr(xp), B, H, H, C, py, py, stream_ptr()))
        r["crop"] = timeit(lambda: call("nnz_crop_top_left", ptr(xp), ptr(xs), B, H, H, C, py, py, stream_ptr()))
        r["ln_f"] = timeit(lambda: call("nnz_layer_norm_forward", ptr(xp), 0, ptr(g), ptr(bta), ptr(yln), 0, ptr(mean), ptr(rstd),
                                        None, Tp, C, 1e-6, stream_ptr()))
        r["ln_b"] = timeit(lambda: call("nnz_layer_norm_backward_det_res", ptr(xp), 0, ptr(g), ptr(mean), ptr(rstd), ptr(yln), 0,
                                        ptr(po), ptr(dx), ptr(dw), ptr(db), ptr(sc.acc), ptr(sc.counter), Tp, C, stream_ptr()))
        r["qkv"] = timeit(lambda: call("nnz_dense32_forward", ptr(yln), ptr(wqkv), ptr(bqkv), ptr(qkv), None, Tp, C, 3 * C, 0,
                                       stream_ptr()))
        r["attn_f"] = timeit(lambda: call("nnz_window_attention_forward", ptr(qkv), ptr(table), ptr(idx), ptr(ao), B, Hp, Hp, C,
                                          heads, 3, (C // heads) ** -0.5, stream_ptr()))
        r["proj"] = timeit(lambda: call("nnz_dense32_forward", ptr(ao), ptr(wp), ptr(bp), ptr(po), None, Tp, C, C, 0, stream_ptr()))
        r["res_f"] = timeit(lambda: call("nnz_residual_droppath_forward", ptr(xp), 0, ptr(po), 0, None, 0, 1.0, ptr(dx), 0, B,
                                         Hp * Hp * C, stream_ptr()))
        r["res_b"] = timeit(lambda: call("nnz_residual_droppath_backward", ptr(po), 0, None, 0, 1.0, ptr(dx), 0, B, Hp * Hp * C,
                                         stream_ptr()))
        r["fc1g"] = timeit(lambda: call("nnz_dense32_forward", ptr(yln), ptr(w1), ptr(b1), ptr(h), ptr(act), Tp, C, 4 * C, 1,
                                        stream_ptr()))
        r["fc2"] = timeit(lambda: call("nnz_dense32_forward", ptr(act), ptr(w2), ptr(b2), ptr(po), None, Tp, 4 * C, C, 0,
                                       stream_ptr()))
        r["fc2_dg"] = timeit(lambda: call("nnz_dense32_dgrad", ptr(po), ptr(w2), ptr(h), ptr(act), Tp, 4 * C, C, stream_ptr()))
        r["fc1_dg"] = timeit(lambda: call("nnz_dense32_dgrad", ptr(act), ptr(w1), None, ptr(dx), Tp, C, 4 * C, stream_ptr()))
        r["proj_dg"] = timeit(lambda: call("nnz_dense32_dgrad", ptr(po), ptr(wp), None, ptr(dx), Tp, C, C, stream_ptr()))
        r["qkv_dg"] = timeit(lambda: call("nnz_dense32_dgrad", ptr(dqkv), ptr(wqkv), None, ptr(dx), Tp, C, 3 * C, stream_ptr()))
        r["attn_b"] = timeit(lambda: call("nnz_window_attention_backward", ptr(qkv), ptr(table), ptr(idx), ptr(ao), ptr(dqkv),
                                          ptr(dtab), ptr(sc.acc), ptr(sc.counter), B, Hp, Hp, C, heads, 3, (C // heads) ** -0.5,
                                          stream_ptr()))
        f = r["pad"] + 2 * r["ln_f"] + r["qkv"] + r["attn_f"] + r["proj"] + 2 * r["res_f"] + r["fc1g"] + r["fc2"] + r["crop"]
        b = r["crop"] + 2 * r["res_b"] + r["fc2_dg"] + r["fc1_dg"] + 2 * r["ln_b"] + r["proj_dg"] + r["attn_b"] + r["qkv_dg"] + \
            r["pad"]
        tot_f += f * blocks
        tot_b += b * blocks
        print(f"{H:4d}^2 C={C:4d} heads={heads:2d} {blocks:3d} " + " ".join(f"{r[n]:7.1f}" for n in names) + f"  {f:7.1f}  {b:7.1f}")
    print(f"sum over the 144 blocks: forward {tot_f / 1e3:.2f} ms, backward (without weight gradients) {tot_b / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
