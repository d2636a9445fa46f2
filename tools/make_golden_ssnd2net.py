"""Round-4 stage-wise fixtures for SSND2Net / SSND2NetP (nets/ssnd2net.py:1143-1405, 1446-1707), generated in the BUILD CONTAINER
from the reference's own classes imported under tools/ref_shim.py (substitutions listed there; the scan is the reference's
selective_scan_ref).  Only arrays / shapes / numbers are stored (tests/golden/), no reference source travels.
    python tools/make_golden_ssnd2net.py [SSND2NetP SSND2Net] [--dims 2 3]

Why stage-wise: the whole network at its seeded initialisation is CHAOTIC - the reference changes its own outputs by 40-96 %
when the input moves by 1e-6 (InstanceNorm / LayerNorm over 2 x 2 ... 4 x 4 maps in ~100 sequential blocks; measured by this
script, `sens` below).  A whole-net output therefore pins nothing.  Instead every top-level module of the network (the eleven
MU stages, patch merging / expanding, the skip-fusion Linears, the six side convolutions, the fuse convolution) is recorded
with the REFERENCE's own input, its output, and the backward of a fixed output gradient: dx and the L2 norm of every parameter
gradient.  The network is built from torch.manual_seed(0) exactly as the trainers do (the product's construction draws the
same parameters bit for bit: tests/golden/seeded_init.json), in eval mode (DropPath off, BatchNorm running statistics).
Stage 1 is the end-to-end piece: its input is the network input itself.

Per module the fixture also holds `sens`: the relative change of the reference's own output when its input is perturbed by
1e-6 (relative): the conditioning the parity test scales its tolerance with (a module that amplifies 1e-6 to more than 5 % is
marked chaotic and compared in structure and scale only).
Large tensors (full-resolution decoder side) are stored as strided samples; inputs are always complete.

Round 5: (1) `bsens` - the REFERENCE's own backward conditioning: the relative change of its dx when its input
moves by 1e-6 (the same perturbation as `sens`); the parity test scales its backward tolerances with it instead of measuring the
product's own response.  (2) The full-resolution decoder-side modules that were "statistics only" are now compared too, on an
input the test can rebuild without storage: every input tensor = cos pattern scaled to the rms of the reference's own activation
at that place (`synth_in_rms`); stored: strided samples of the output and of dx, the norms, every parameter-gradient norm, `sens`
and `bsens` on that input (`synth_*` keys; all earlier arrays are unchanged and regenerate bit for bit)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
MAX_IO_FLOATS = 420_000      # modules whose inputs + output are larger are recorded with a strided output and no backward


def pattern(shape, freq, phase):
    i = torch.arange(int(np.prod(shape)), dtype=torch.float64)
    return torch.cos(freq * i + phase).float().reshape(shape)


def synthetic_record(mod, name, ins, kwargs, rec, arrays):
    """forward / backward of `mod` on formula-made inputs scaled to the rms of the reference's activations `ins`"""
    rms = [float(t.double().pow(2).mean().sqrt()) for t in ins]
    syn = [r * pattern(t.shape, 0.61 + 0.13 * k, 0.3) for k, (t, r) in enumerate(zip(ins, rms))]
    rec["synth_in_rms"] = rms
    xin = [t.clone().requires_grad_(True) for t in syn]
    for p in mod.parameters():
        p.grad = None
    y = mod(*xin, **kwargs)
    y.backward(pattern(y.shape, 0.37, 0.5))
    out, dx = y.detach(), xin[0].grad
    rec["synth_out_stride"] = int(max(1, out.numel() // 65536))
    rec["synth_dx_stride"] = int(max(1, dx.numel() // 16384))
    arrays[f"synth_out_{name}"] = out.reshape(-1)[::rec["synth_out_stride"]].numpy().copy()
    arrays[f"synth_dx_{name}"] = dx.reshape(-1)[::rec["synth_dx_stride"]].numpy().copy()
    rec["synth_out_rms"] = float(out.double().pow(2).mean().sqrt())
    rec["synth_out_max"] = float(out.abs().max())
    rec["synth_dx_norm"] = float(dx.double().pow(2).sum().sqrt())
    rec["synth_dx_max"] = float(dx.abs().max())
    names, norms = [], []
    for pn, p in mod.named_parameters():
        if p.grad is not None:
            names.append(pn)
            norms.append(float(p.grad.double().pow(2).sum().sqrt()))
    rec["synth_grad_names"] = names
    arrays[f"synth_gn_{name}"] = np.array(norms, dtype=np.float64)
    for p in mod.parameters():
        p.grad = None
    pert = [syn[0] + 1e-6 * rms[0] * pattern(syn[0].shape, 1.3, 0.2)] + syn[1:]
    xin2 = [t.clone().requires_grad_(True) for t in pert]
    y2 = mod(*xin2, **kwargs)
    y2.backward(pattern(y2.shape, 0.37, 0.5))
    rec["synth_sens"] = float((y2.detach() - out).abs().max() / out.abs().max())
    rec["synth_bsens"] = float((xin2[0].grad - dx).abs().max() / dx.abs().max())
    for p in mod.parameters():
        p.grad = None


def run_case(R, cls, sd, P, out_prefix):
    torch.manual_seed(0)
    net = getattr(R, cls)(spatial_dims=sd, factorization_type="cross-scan", in_ch=1, out_ch=2, deep_supervision=True,
                          input_patch_size=[P] * sd)
    net.eval()
    x = torch.randn(1, 1, *([P] * sd), generator=torch.Generator().manual_seed(5))
    calls = []

    def hook(name):
        def fn(mod, args, kwargs, out):
            calls.append((name, [a.detach().clone() for a in args if torch.is_tensor(a)], dict(kwargs), out.detach().clone()))
        return fn

    handles = [m.register_forward_hook(hook(n), with_kwargs=True) for n, m in net.named_children()]
    t0 = time.time()
    with torch.no_grad():
        outs = net(x)
    for h in handles:
        h.remove()
    print(cls, sd, "forward", round(time.time() - t0, 1), "s,", len(calls), "module calls", flush=True)
    arrays = {"x": x.numpy()}
    man = {"cls": cls, "spatial_dims": sd, "patch": P, "modules": []}
    for i, o in enumerate(outs):
        arrays[f"netout{i}_rms"] = np.array([float(o.double().pow(2).mean().sqrt())])
    for name, ins, kwargs, out in calls:
        mod = getattr(net, name)
        nfl = sum(t.numel() for t in ins) + out.numel()
        small = nfl <= MAX_IO_FLOATS
        rec = {"name": name, "kwargs": kwargs, "in_shapes": [list(t.shape) for t in ins], "out_shape": list(out.shape),
               "full": bool(small), "params": sum(p.numel() for p in mod.parameters())}
        # conditioning: relative output change for a 1e-6 relative perturbation of the (first) input
        with torch.no_grad():
            scale = float(ins[0].double().pow(2).mean().sqrt())
            pert = [ins[0] + 1e-6 * scale * pattern(ins[0].shape, 1.3, 0.2)] + ins[1:]
            o2 = mod(*pert, **kwargs)
            rec["sens"] = float((o2 - out).abs().max() / out.abs().max())
        if not small and name != "stage1":
            # large decoder-side modules: the reference's activations are too large to store - output statistics of the real call,
            # and a full forward / backward record on a formula-made input of the same scale
            rec["out_rms"] = float(out.double().pow(2).mean().sqrt())
            synthetic_record(mod, name, ins, kwargs, rec, arrays)
            man["modules"].append(rec)
            print(f"  {name:20s} large ({nfl * 4 // 1024} KB): formula-made input, sens {rec['synth_sens']:.2e} "
                  f"bsens {rec['synth_bsens']:.2e}", flush=True)
            continue
        for k, t in enumerate(ins):
            if name == "stage1":
                continue                     # its input is x
            arrays[f"in{k}_{name}"] = t.numpy()
        if small:
            arrays[f"out_{name}"] = out.numpy()
            rec["out_stride"] = 1
        else:
            flat = out.reshape(-1)
            rec["out_stride"] = int(max(1, flat.numel() // 65536))
            arrays[f"out_{name}"] = flat[::rec["out_stride"]].numpy().copy()
        rec["out_rms"] = float(out.double().pow(2).mean().sqrt())
        # backward of a fixed output gradient through the module alone
        xin = [t.clone().requires_grad_(True) for t in ins]
        for p in mod.parameters():
            p.grad = None
        y = mod(*xin, **kwargs)
        dy = pattern(y.shape, 0.37, 0.5)
        y.backward(dy)
        dx = xin[0].grad
        flat = dx.reshape(-1)
        rec["dx_stride"] = int(max(1, flat.numel() // 16384))
        arrays[f"dx_{name}"] = flat[::rec["dx_stride"]].numpy().copy()
        rec["dx_norm"] = float(dx.double().pow(2).sum().sqrt())
        names, norms = [], []
        for pn, p in mod.named_parameters():
            if p.grad is not None:
                names.append(pn)
                norms.append(float(p.grad.double().pow(2).sum().sqrt()))
        rec["grad_names"] = names
        arrays[f"gn_{name}"] = np.array(norms, dtype=np.float64)
        for p in mod.parameters():
            p.grad = None
        # the reference's own backward conditioning: dx of the same output gradient at the perturbed input
        xin2 = [pert[0].clone().requires_grad_(True)] + [t.clone().requires_grad_(True) for t in ins[1:]]
        y2 = mod(*xin2, **kwargs)
        y2.backward(pattern(y2.shape, 0.37, 0.5))
        rec["bsens"] = float((xin2[0].grad - dx).abs().max() / dx.abs().max())
        for p in mod.parameters():
            p.grad = None
        man["modules"].append(rec)
        print(f"  {name:20s} sens {rec['sens']:.2e}  bsens {rec['bsens']:.2e}  out_rms {rec['out_rms']:.3e}  dx_norm {rec['dx_norm']:.3e}  "
              f"{len(names)} parameter gradients", flush=True)
    np.savez_compressed(out_prefix + ".npz", **arrays)
    with open(out_prefix + ".json", "w") as f:
        json.dump(man, f)
    print("wrote", out_prefix, round(os.path.getsize(out_prefix + ".npz") / 2 ** 20, 2), "MB", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("classes", nargs="*", default=["SSND2NetP", "SSND2Net"])
    ap.add_argument("--dims", nargs="*", type=int, default=[2, 3])
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--max-io-floats", type=int, default=MAX_IO_FLOATS)
    a = ap.parse_args()
    globals()["MAX_IO_FLOATS"] = a.max_io_floats
    torch.set_num_threads(a.threads)
    ref_shim.install()
    from nnunetv2.nets import ssnd2net as R
    for cls in a.classes:
        for sd in a.dims:
            P = {2: 96, 3: 24}[sd]
            run_case(R, cls, sd, P, os.path.join(OUT, f"stages_{cls}_{sd}d"))


if __name__ == "__main__":
    main()
