"""Where does a conv_box workgroup spend its time?  Phase timestamps (s_memtime of thread 0) of every workgroup of one launch.

    python tools/probes/conv_phase_probe.py --build          # here or on the GPU box: the instrumented library (never the shipped one)
    python tools/probes/conv_phase_probe.py [--only enc0.1]  # GPU box

Builds tools/probes/_ts/libnnuzoo_hip_ts.so = the product objects with csrc/conv_fprop.hip recompiled under
-DNNZ_CONV_TIMESTAMPS=1, loads it INSTEAD of libnnuzoo_hip.so in this process only, and runs single launches as the training step
issues them (forward with the consumer-side norm and the statistics epilogue; data gradient with the fused norm-backward
reductions; both also plain).  Slots (conv_fprop.hip NNZ_TS):
0 entry, 1 set-up done, 2 + 2k slice k staged in LDS (both barriers passed), 3 + 2k slice k's MFMA loop done, 12 accumulators
transposed into LDS, 13 statistics / reductions done, 14 stores issued.  Printed: mean cycles per phase over the workgroups, the
share of the workgroup's lifetime, and the launch's wall time against (workgroups / 512 resident) x mean lifetime."""
import argparse
import time
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
TS_DIR = os.path.join(ROOT, "tools", "probes", "_ts")
TS_LIB = os.path.join(TS_DIR, "libnnuzoo_hip_ts.so")


def build_wgrad():
    """the product objects + csrc/conv_wgrad.hip under -DNNZ_WGRAD_TIMESTAMPS=1 -> tools/probes/_ts/libnnuzoo_hip_wts.so"""
    from nnuzoo_amd import build as B
    B.build(verbose=False)
    os.makedirs(TS_DIR, exist_ok=True)
    obj = os.path.join(TS_DIR, "wts_conv_wgrad.o")
    subprocess.run([B.HIPCC, *B._flags("conv_wgrad.hip"), "-DNNZ_WGRAD_TIMESTAMPS=1", "-c", os.path.join(B.CSRC, "conv_wgrad.hip"),
                    "-o", obj], check=True)
    objs = [os.path.join(B.OBJ, s.replace(".hip", ".o")) for s in B._sources() if s != "conv_wgrad.hip"] + [obj]
    lib = os.path.join(TS_DIR, "libnnuzoo_hip_wts.so")
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
    print("built", lib)
    return lib


def wgrad_phases(a):
    """per-phase cycle totals of the weight-gradient workgroups (thread 0, summed over the workgroup's tiles)"""
    lib = os.path.join(TS_DIR, "libnnuzoo_hip_wts.so")
    if not os.path.exists(lib):
        build_wgrad()
    import torch
    from nnuzoo_amd import _lib
    _lib.LIB_PATH = lib
    from nnuzoo_amd import conv_plan as cp
    from nnuzoo_amd import hip_ops as ops
    from nnuzoo_amd.hip_ops import PreparedTable
    dev = torch.device("cuda")
    N = 2
    ts = torch.zeros(1 << 14, 8, dtype=torch.int64, device=dev)
    addr = ts.data_ptr()
    sweeps = [()]
    if a.tuning:      # 'k=v,k=v;k=v,...': one table per ';'-separated setting
        sweeps = [tuple((int(kv.split("=")[0]), int(kv.split("=")[1])) for kv in st.split(",") if kv) for st in a.tuning.split(";")]

    def arm(on):
        lo, hi = (addr & 0xFFFFFFFF, addr >> 32) if on else (0, 0)
        _lib.call("nnz_conv_tuning", 12, lo - (1 << 32) if lo >= (1 << 31) else lo)
        _lib.call("nnz_conv_tuning", 13, hi)

    layers = [("enc0.1", 32, 32, 128, 1), ("dec0.0", 64, 32, 128, 1), ("enc1.1", 64, 64, 64, 1), ("dec1.0", 128, 64, 64, 1),
              ("enc2.1", 128, 128, 32, 1), ("enc1.0", 32, 64, 128, 2)]
    names = ["set-up + first loads issued", "wait + stage (2 barriers)", "issue next tile's loads", "MFMA loop", "flush"]
    for name, cin, cout, edge, stride in layers:
        if a.only and a.only not in name:
            continue
        dims = (edge,) * 3
        od = cp.conv_out_dims(dims, (3, 3, 3), stride)
        V, Vo = edge ** 3, int(np.prod(od))
        x = torch.randn(N, V, cin, device=dev).to(torch.float16)
        dy = torch.randn(N, Vo, cout, device=dev).to(torch.float16)
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        gw = torch.empty_like(w)
        tab = torch.randn(N, cin, 4, device=dev)
        tab[:, :, 2] = 1.0 + 0.1 * tab[:, :, 2]
        inn = ops.InNorm(tab, 0.01)
        pw = PreparedTable(cp.conv_wgrad(N, dims, cin, cout, stride=stride))
        ws = torch.empty(ops.conv_tap_wgrad_workspace_floats(pw), device=dev, dtype=torch.float32)
        runs = {"plain": lambda: ops.conv_tap_wgrad_to_grad(pw, x, dy, ws, gw, 27, cin * 27, 1)}
        if stride == 1:
            pwf = PreparedTable(cp.conv_wgrad_flipped(N, dims, cin, cout))
            runs["step form (flipped, plain operand normalised)"] = \
                lambda: ops.conv_tap_wgrad_to_grad(pwf, dy, x, ws, gw, cin * 27, 27, 1, plain_norm=inn)
        else:
            runs["step form (boxed operand normalised)"] = lambda: ops.conv_tap_wgrad_to_grad(pw, x, dy, ws, gw, 27, cin * 27, 1, boxed_norm=inn)
        for what, fn, sweep in [(w_, f_, s_) for w_, f_ in runs.items() for s_ in sweeps]:
            for k, v in sweep:
                _lib.call("nnz_conv_tuning", k, v)
            if sweep:
                what = f"{what} [knobs {sweep}]"
            arm(False)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for _ in range(5):
                fn()
            e0.record()
            torch.cuda.synchronize()
            wall = s0.elapsed_time(e0) / 5 * 1e3
            if a.clock:     # DVFS steady state: back-to-back launches for a.clock seconds before the stamped one
                t_end = time.time() + a.clock
                while time.time() < t_end:
                    for _ in range(50):
                        fn()
                    torch.cuda.synchronize()
            ts.zero_()
            arm(True)
            fn()
            torch.cuda.synchronize()
            arm(False)
            t = ts.cpu().numpy().astype(np.float64)
            t = t[t[:, 5] > 0]
            tot = t[:, :5].sum(1)
            print(f"\n{name} {cin}->{cout} @{edge} s{stride}  wgrad {what}: {len(t)} workgroups x {t[:, 5].mean():.1f} tiles, wall {wall:.1f} us "
                  f"(kernel + fold), mean workgroup lifetime {tot.mean():.0f} ticks")
            for k, nm in enumerate(names):
                per = t[:, k].mean() / (t[:, 5].mean() if k in (1, 2, 3) else 1.0)
                print(f"   {nm:30s} {t[:, k].mean():10.0f} ticks  {100 * t[:, k].mean() / tot.mean():5.1f} %"
                      + (f"   ({per:.0f} per tile)" if k in (1, 2, 3) else ""))
            if a.residency:
                # which workgroups shared a CU: (XCC, SE, SH, CU) from the hardware-id registers the workgroup read at its end
                raw = ts.cpu().numpy()
                live = np.nonzero(raw[:, 5] > 0)[0]
                hw = raw[live, 6].astype(np.uint64)
                xcc, hwid = (hw >> np.uint64(32)) & np.uint64(0xF), hw & np.uint64(0xFFFFFFFF)
                cu = (xcc << np.uint64(8)) | ((hwid >> np.uint64(8)) & np.uint64(0xFF))
                groups = {}
                for b, c in zip(live, cu):
                    groups.setdefault(int(c), []).append(int(b))
                sizes = np.bincount([len(v) for v in groups.values()])
                diffs = np.bincount([abs(v[1] - v[0]) for v in groups.values() if len(v) == 2])
                top = np.argsort(-diffs)[:4]
                print(f"   residency: {len(groups)} distinct CUs; workgroups per CU histogram {dict(enumerate(sizes.tolist()))}; "
                      f"blockIdx distance of the pairs: {[(int(d), int(diffs[d])) for d in top if diffs[d]]}")
                pass
            real = ts.cpu().numpy()[:, 7].astype(np.float64)[ts.cpu().numpy()[:, 5] > 0]
            print(f"   in-kernel shader clock (phase ticks / lifetime on the 100 MHz clock, median over workgroups): "
                  f"{np.median(tot / real * 0.1):.2f} GHz")
            for k, v in sweep:
                _lib.call("nnz_conv_tuning", k, 0)


def build(defines=("-DNNZ_CONV_TIMESTAMPS=1",), lib=TS_LIB):
    """the product objects + csrc/conv_fprop.hip recompiled with `defines` -> `lib` (experiment builds; NNZ_HIP_LIBRARY=<lib> makes
    any tool of the repo load it)"""
    from nnuzoo_amd import build as B
    B.build(verbose=False)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    obj = lib[:-3] + "_conv_fprop.o"
    cmd = [B.HIPCC, *B._flags("conv_fprop.hip"), *defines, "-c", os.path.join(B.CSRC, "conv_fprop.hip"), "-o", obj]
    subprocess.run(cmd, check=True)
    objs = [os.path.join(B.OBJ, s.replace(".hip", ".o")) for s in B._sources() if s != "conv_fprop.hip"] + [obj]
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
    print("built", lib)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--wgrad", action="store_true", help="phase totals of the weight-gradient kernel (-DNNZ_WGRAD_TIMESTAMPS build)")
    ap.add_argument("--build-variant", default="", help="name=-DFLAG[,-DFLAG...]: tools/probes/_ts/libnnuzoo_hip_<name>.so with these defines")
    ap.add_argument("--only", default="")
    ap.add_argument("--tuning", default="", help="k=v[,k=v]: conv tuning knobs; --wgrad: ';' separates settings that are measured in turn")
    ap.add_argument("--clock", type=float, default=0.0, help="seconds of back-to-back launches before the stamped one (DVFS steady state); "
                    "the in-kernel clock is printed either way")
    ap.add_argument("--residency", action="store_true", help="--wgrad: report which workgroups shared a CU")
    a = ap.parse_args()
    if a.wgrad:
        if a.build:
            build_wgrad()
        else:
            wgrad_phases(a)
        return
    if a.build_variant:
        name, flags = a.build_variant.split("=", 1)
        build(tuple(flags.split(",")), os.path.join(TS_DIR, f"libnnuzoo_hip_{name}.so"))
        return
    if a.build:
        build()
        return
    if not os.path.exists(TS_LIB):
        build()
    import torch
    from nnuzoo_amd import _lib
    _lib.LIB_PATH = TS_LIB
    from nnuzoo_amd import conv_plan as cp
    from nnuzoo_amd import hip_ops as ops
    from nnuzoo_amd.hip_ops import PreparedTable
    for kv in filter(None, a.tuning.split(",")):
        k, v = kv.split("=")
        _lib.call("nnz_conv_tuning", int(k), int(v))
    dev = torch.device("cuda")
    N = 2
    ts = torch.zeros(1 << 16, 16, dtype=torch.int64, device=dev)
    addr = ts.data_ptr()

    def arm(on):
        lo, hi = (addr & 0xFFFFFFFF, addr >> 32) if on else (0, 0)
        _lib.call("nnz_conv_tuning", 12, lo - (1 << 32) if lo >= (1 << 31) else lo)
        _lib.call("nnz_conv_tuning", 13, hi)

    names = {0: "entry->setup", 1: "setup->slice0 staged"}
    layers = [("enc0.1", 32, 32, 128, 1), ("dec0.0", 64, 32, 128, 1), ("enc1.1", 64, 64, 64, 1), ("dec1.0", 128, 64, 64, 1),
              ("enc1.0", 32, 64, 128, 2), ("enc2.1", 128, 128, 32, 1)]
    for name, cin, cout, edge, stride in layers:
        if a.only and a.only not in name:
            continue
        dims = (edge,) * 3
        od = cp.conv_out_dims(dims, (3, 3, 3), stride)
        V, Vo = edge ** 3, int(np.prod(od))
        x = torch.randn(N, V, cin, device=dev).to(torch.float16)
        dy = torch.randn(N, Vo, cout, device=dev).to(torch.float16)
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        y = torch.empty(N, Vo, cout, device=dev, dtype=torch.float16)
        dx = torch.empty(N, V, cin, device=dev, dtype=torch.float16)
        pf = PreparedTable(cp.conv_forward(N, dims, cin, cout, stride=stride))
        pd = PreparedTable(cp.conv_dgrad(N, dims, cin, cout, stride=stride))
        wf = ops.pack_weight(w, pf, cin, cout, 27, cin * 27, 1)
        wd = ops.pack_weight(w, pd, cout, cin, cin * 27, 27, 1)
        tab = torch.randn(N, cin, 4, device=dev)
        tab[:, :, 2] = 1.0 + 0.1 * tab[:, :, 2]
        inn = ops.InNorm(tab, 0.01)
        sc = ops.NormScratch(dev, N * max(cin, cout))
        gamma, beta = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        nstat = torch.empty(N, cout, 4, device=dev)
        xstat = torch.randn(N, cin, 4, device=dev)
        nred, dgam, dbet = torch.empty(N, cin, 2, device=dev), torch.empty(cin, device=dev), torch.empty(cin, device=dev)
        runs = {"fwd+innorm+stats": lambda: ops.conv_tap_forward_norm(pf, x, wf, None, y, sc, gamma, beta, 1e-5, nstat, innorm=inn),
                "fwd plain": lambda: ops.conv_tap_forward(pf, x, wf, None, y),
                "dgrad+normred": lambda: ops.conv_tap_dgrad_normred(pd, dy, wd, dx, x, cin, xstat, 0.01, sc, nred, dgam, dbet),
                "dgrad plain": lambda: ops.conv_tap_forward(pd, dy, wd, None, dx)}
        for what, fn in runs.items():
            arm(False)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                fn()
            e.record()
            torch.cuda.synchronize()
            wall = s.elapsed_time(e) / 5 * 1e3          # us, uninstrumented pointer (the branch is still compiled in)
            if a.clock:     # DVFS steady state: back-to-back launches for a.clock seconds before the stamped one
                t_end = time.time() + a.clock
                while time.time() < t_end:
                    for _ in range(50):
                        fn()
                    torch.cuda.synchronize()
            ts.zero_()
            arm(True)
            fn()
            torch.cuda.synchronize()
            arm(False)
            t = ts.cpu().numpy().astype(np.int64)
            live = t[:, 0] != 0
            t = t[live]
            nwg = len(t)
            if (t[:, 15] > 0).all() and cin <= 32:   # <= 2 slices: slot 15 = lifetime on the constant 100 MHz clock
                clk = np.median((t[:, 14] - t[:, 0]) / t[:, 15].astype(np.float64) * 0.1)
                print(f"   [{what}] in-kernel shader clock (s_memtime ticks / 100 MHz lifetime, median over workgroups): {clk:.2f} GHz")
            life = (t[:, 14] - t[:, 0]).astype(np.float64)     # (the XCDs' counters are not synchronised: only differences
            used = [s_ for s_ in range(15) if (t[:, s_] != 0).all()]   #  inside one workgroup mean anything)
            used.sort(key=lambda s_: float((t[:, s_] - t[:, 0]).mean()))   # time order (slots 6..11 double as epilogue sub-phases)
            print(f"\n{name} {cin}->{cout} @{edge} s{stride}  {what}: {nwg} workgroups, wall {wall:.1f} us, mean workgroup lifetime "
                  f"{life.mean():.0f} ticks; {nwg / 512:.1f} rounds of 512 resident workgroups -> {wall / (nwg / 512) :.2f} us per round")
            prev = used[0]
            for s_ in used[1:]:
                d = (t[:, s_] - t[:, prev]).astype(np.float64)
                before_epilogue = 12 in used and used.index(s_) < used.index(12) and 9 not in used[:used.index(s_) + 1]
                sub = {6: "stats: local sums", 7: "stats: lane fold + slab + barrier", 8: "stats: wave 0 fixed-point adds",
                       9: "last loop -> layer-below loads issued", 10: "normred: local sums (+ stores)",
                       11: "normred: lane fold + slab + barrier", 12: "last loop -> acc in LDS image", 13: "statistics / wave-0 adds",
                       14: "stores / tail"}
                if s_ == 1:
                    label = "entry -> set-up done"
                elif s_ == 2:
                    label = "set-up -> slice 0 staged"
                elif before_epilogue:
                    label = f"slice {(s_ - 3) // 2} MFMA loop" if s_ % 2 == 1 else f"slice {(s_ - 2) // 2} wait + stage"
                else:
                    label = sub.get(s_, f"{prev}->{s_}")
                print(f"   {prev:2d}->{s_:2d}  {label:38s} mean {d.mean():9.0f}  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f}"
                      f"   {100 * d.mean() / life.mean():5.1f} %")
                prev = s_


if __name__ == "__main__":
    main()
