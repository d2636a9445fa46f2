"""Sliding-window inference throughput of the 3d_fullres nnUNet on one MI355X (the reference measures
this path with run_test.py / nnUNetPredictor).  Synthetic image, random-init network of the planner's shape.

    python tools/bench_inference.py [--edge 256] [--patch 128] [--no-mirror] [--reps 2]
prints tiles/s (network forwards incl. mirror variants), image voxels/s and the share of the accumulation kernels.
"""
import argparse
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nnuzoo_amd.inference.predict_from_raw_data import nnUNetPredictor  # noqa: E402
from nnuzoo_amd.synthetic import nnunet_plans  # noqa: E402
from nnuzoo_amd.utilities.get_network_from_plans import get_network_from_plans  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edge", type=int, default=256)
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--no-mirror", action="store_true")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--tiles-per-forward", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda")
    plans, cfg, dj = nnunet_plans(3, (a.patch,) * 3, batch_size=2)
    arch = plans["configurations"][cfg]["architecture"]
    net = get_network_from_plans(arch["network_class_name"], arch["arch_kwargs"], arch["_kw_requires_import"], 1, 2,
                                 allow_init=True, deep_supervision=False).to(dev).eval()
    pr = nnUNetPredictor(tile_step_size=0.5, use_gaussian=True, use_mirroring=not a.no_mirror, device=dev,
                         allow_tqdm=False, tiles_per_forward=a.tiles_per_forward)
    pr.manual_initialization(net, None, types.SimpleNamespace(patch_size=[a.patch] * 3), None, dj, "nnUNetTrainer",
                             None if a.no_mirror else (0, 1, 2),
                             label_manager=types.SimpleNamespace(num_segmentation_heads=2))
    img = torch.randn(1, a.edge, a.edge, a.edge, device=dev)
    nt = len(pr._internal_get_sliding_window_slicers(img.shape[1:]))
    m = 1 if a.no_mirror else 8
    pr.predict_sliding_window_return_logits(img)  # warm-up (plans, packed weights, gaussian)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = pr.predict_sliding_window_return_logits(img)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    print(f"image {a.edge}^3, patch {a.patch}^3, {nt} tiles x {m} mirror variants: {dt * 1e3:.1f} ms per image, "
          f"{nt * m / dt:.1f} tile forwards/s, {a.edge ** 3 / dt / 1e6:.1f} Mvoxel/s, peak mem "
          f"{torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB, finite={bool(torch.isfinite(out.float()).all())}")


if __name__ == "__main__":
    main()
