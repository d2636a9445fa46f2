"""Training throughput of the 2-D nnUNet (BASELINE configs[0]: nnUNet 2d, synthetic 1x512x512 patches) on one MI355X.
The reference runs this configuration on CPU as its plumbing check; here it goes through the same HIP schedule as the
3-D network (2-D layers are depth-1 volumes).  python tools/bench_2d.py [--batch 12] [--size 512] [--steps 10]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch  # noqa: E402
from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    plans, cfg, dj = nnunet_plans(2, (a.size, a.size), batch_size=a.batch)
    tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    batch = synthetic_batch(a.batch, (a.size, a.size), tr._get_deep_supervision_scales(), seed=1)
    batch = {"data": batch["data"].cuda(), "target": [t.cuda() for t in batch["target"]]}
    losses = []
    for _ in range(a.warmup):
        losses.append(float(tr.train_step(batch)["loss"]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses.append(float(tr.train_step(batch)["loss"]))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    arch = plans["configurations"][cfg]["architecture"]["arch_kwargs"]
    print(json.dumps({"model": "nnUNet 2d (PlainConvUNet, %d stages)" % arch["n_stages"], "patch": a.size,
                      "batch": a.batch, "ms_per_step": round(dt * 1e3, 2), "patches_per_s": round(a.batch / dt, 1),
                      "loss_first_last": [round(losses[0], 4), round(losses[-1], 4)],
                      "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
