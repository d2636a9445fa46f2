"""Where does a workgroup of win_attn_bwd_pair_kernel spend its time?  s_memtime of thread 0 at the phase boundaries of every
workgroup of one launch (library rebuilt with -DNNZ_WA_TIMESTAMPS=1 into tools/probes/_ts/, never the shipped one).
Slots: 0 entry, 1 set-up done, 2 bias matrix in LDS, 3 first window's images staged, 4 pass A done, 5 K / V rows taken
(pass B starts), 6 all windows done, 7 bias-gradient shares written (end).
    python tools/probes/wa_phase_probe.py            # GPU box (builds the library first if it is missing)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
TS_DIR = os.path.join(ROOT, "tools", "probes", "_ts")
LIB = os.path.join(TS_DIR, "libnnuzoo_hip_wats.so")


def build():
    from nnuzoo_amd import build as B
    B.build(verbose=False)
    os.makedirs(TS_DIR, exist_ok=True)
    obj = os.path.join(TS_DIR, "wats_window_attention.o")
    subprocess.run([B.HIPCC, *B._flags("window_attention.hip"), "-DNNZ_WA_TIMESTAMPS=1", "-c",
                    os.path.join(B.CSRC, "window_attention.hip"), "-o", obj], check=True)
    objs = [os.path.join(B.OBJ, s.replace(".hip", ".o")) for s in B._sources() if s != "window_attention.hip"] + [obj]
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], check=True)
    print("built", LIB)


def main():
    if "--build" in sys.argv or not os.path.exists(LIB):
        build()
        if "--build" in sys.argv:
            return
    import torch
    from nnuzoo_amd import _lib
    _lib.LIB_PATH = LIB
    from nnuzoo_amd._lib import call, ptr, stream_ptr
    from nnuzoo_amd.hip_ops import det_scratch
    raw = C.CDLL(LIB)
    raw.wa_probe_set_timestamps.argtypes = [C.c_void_p]
    ar = torch.arange(7)
    yy, xx = torch.meshgrid(ar, ar, indexing="ij")
    y, x = yy.flatten(), xx.flatten()
    idx = ((y[:, None] - y[None, :] + 6) * 13 + (x[:, None] - x[None, :] + 6)).to(torch.int32).cuda()
    names = ["set-up", "bias matrix -> LDS", "token map + images of window 0", "pass A", "hand-over to pass B", "pass B + further windows",
             "bias-gradient shares"]
    for H, heads in [(7, 24), (14, 24), (35, 12), (133, 3)]:
        B, hd = 2, 32
        Cc = heads * hd
        qkv = torch.randn(B, H, H, 3 * Cc, device="cuda")
        table = torch.randn(169, heads, device="cuda") * 0.5
        dout = torch.randn(B, H, H, Cc, device="cuda")
        dqkv, dtable = torch.empty_like(qkv), torch.empty_like(table)
        sc = det_scratch(qkv.device, 170 * heads)
        ts = torch.zeros(1 << 16, 16, dtype=torch.int64, device="cuda")

        def run():
            call("nnz_window_attention_backward", ptr(qkv), ptr(table), ptr(idx), ptr(dout), ptr(dqkv), ptr(dtable), ptr(sc.acc),
                 ptr(sc.counter), B, H, H, Cc, heads, 0, hd ** -0.5, stream_ptr())
        raw.wa_probe_set_timestamps(None)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            run()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        raw.wa_probe_set_timestamps(C.c_void_p(ts.data_ptr()))
        run()
        torch.cuda.synchronize()
        raw.wa_probe_set_timestamps(None)
        t = ts.cpu().numpy()
        live = t[:, 7] > 0
        t = t[live]
        d = np.diff(t[:, :8].astype(np.int64), axis=1)
        life = (t[:, 7] - t[:, 0]).mean()
        span = (t[:, 7].max() - t[:, 0].min())
        print(f"{H}^2 x {heads} heads: {2 * (H // 7) ** 2 * heads} (window, head), {live.sum()} workgroups, launch {us:.1f} us; s_memtime (shader cycles):"
              f" mean workgroup lifetime {life:.0f}, first entry -> last exit {span:.0f} cycles")
        for k, nme in enumerate(names):
            print(f"    {nme:34s} {d[:, k].mean():8.0f} cycles  {100 * d[:, k].mean() / life:5.1f} %")


if __name__ == "__main__":
    main()
