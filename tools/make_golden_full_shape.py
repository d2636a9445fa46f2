"""Full-shape forward fixture of the SS2D^2Net configuration (BASELINE configs[2]: M2Net, 1 x 512^2), VERDICT r5 item 6.
Run ONCE in the build container (about an hour of CPU: the oracle's selective scan is a Python loop over the 512^2 / 4 ... tokens of
every SS2D block):
    python tools/make_golden_full_shape.py [--size 512] [--net M2Net]
What runs is oracle/m2net.py - the CPU restatement of /root/reference/nnunetv2/nets/m2net.py:883-956 that tests/test_oracle_m2net.py
pins to the reference's own whole-net outputs and autograd - on golden_util.det_fill parameters (a formula, no RNG) and the seeded
synthetic input, eval mode.  Stored (data only): strided samples of the seven outputs, their shapes / ranges / L2 norms, the oracle's own
relative response to a 1e-6 input perturbation per output (`sens`: the yardstick the parity test scales its tolerance with), the
packed argmax mask and the top-2 margin (fp16) of the full-resolution output, L2 norms of the parameters and the input (so that the
test can tell a wrong fill from a wrong forward)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from golden_util import det_fill  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="M2Net")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--samples", type=int, default=16384)
    ap.add_argument("--sens", type=int, default=0, help="1: a second forward on a 1e-6 perturbed input (doubles the run time)")
    a = ap.parse_args()
    from oracle import m2net as om
    from nnuzoo_amd.synthetic import synthetic_batch
    torch.manual_seed(0)
    ref = getattr(om, a.net)(1, 2, True)
    det_fill(ref)
    for m in ref.modules():
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
    ref.eval()
    x = synthetic_batch(1, (a.size, a.size), [[1, 1]], seed=a.seed)["data"]
    t = time.time()
    with torch.no_grad():
        base = ref(x)
        print(f"forward {time.time() - t:.0f} s", flush=True)
        pert = ref(x * (1 + 1e-6)) if a.sens else base     # (eval-mode M2Net answers 1e-6 with 2e-6 ... 5e-6: test_m2net_vs_oracle_gpu.py)
    print(f"both forwards {time.time() - t:.0f} s", flush=True)
    out = {"seed": np.int64(a.seed), "size": np.int64(a.size),
           "param_l2": np.float64(float(sum(p.double().pow(2).sum() for p in ref.parameters()).sqrt())),
           "input_l2": np.float64(float(x.double().pow(2).sum().sqrt()))}
    for i, (r, p) in enumerate(zip(base, pert)):
        stride = max(1, r.numel() // a.samples)
        rng = r.abs().max().item()
        out[f"shape_{i}"] = np.array(r.shape, dtype=np.int64)
        out[f"stride_{i}"] = np.int64(stride)
        out[f"out_{i}"] = r.reshape(-1)[::stride].numpy().astype(np.float32)
        out[f"range_{i}"] = np.float64(rng)
        out[f"l2_{i}"] = np.float64(float(r.double().pow(2).sum().sqrt()))
        out[f"sens_{i}"] = np.float64((p - r).abs().max().item() / rng)
        print(i, tuple(r.shape), "range", rng, "sens", out[f"sens_{i}"], flush=True)
    d0 = base[0][0]
    top2 = d0.topk(2, dim=0).values
    out["argmax_0"] = np.packbits(d0.argmax(0).numpy().astype(np.uint8).reshape(-1))
    out["margin_0"] = (top2[0] - top2[1]).numpy().astype(np.float16).reshape(-1)
    path = os.path.join(ROOT, "tests", "golden", f"full_shape_{a.net}_{a.size}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
