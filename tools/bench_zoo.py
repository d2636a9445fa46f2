"""Training patches/s of the zoo models on synthetic 1x512^2 patches (BASELINE.json configs[2], configs[3]).
Usage (GPU box): python tools/bench_zoo.py [--models M2Net,SwT2Net] [--batch 2] [--steps 5] [--size 512]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training import zoo_trainers as Z


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="M2NetP,M2Net,SwT2Net")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--graph", type=int, default=1)  # 1: hipGraph replay of fwd+loss+bwd (the trainers' default), 0: eager
    a = ap.parse_args()
    for name in a.models.split(","):
        cls = getattr(Z, "nnUNetTrainer" + name)
        plans, cfg, dj = nnunet_plans(2, (a.size, a.size), batch_size=a.batch)
        torch.manual_seed(0)
        tr = cls(plans, cfg, 0, dj, device=torch.device("cuda"))
        tr.initialize()
        tr.use_hip_graph = bool(a.graph)
        scales = tr._get_deep_supervision_scales()
        b = synthetic_batch(a.batch, (a.size, a.size), scales if scales is not None else [[1.0, 1.0]], seed=3)
        # single-output trainers (deep supervision off: SegMamba, LightMUNet, SwinTransformerUnet) take one target tensor
        b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]] if scales is not None else b["target"][0].cuda()}
        losses = []
        for _ in range(a.warmup):
            losses.append(float(tr.train_step(b)["loss"]))
        torch.cuda.synchronize()
        # GradScaler bookkeeping of the timed window: a step whose fp16 backward overflowed is SKIPPED and halves the loss
        # scale (growth: one doubling per 2 000 applied steps - none inside a bench window), so log2(scale before / after) counts
        # the skipped steps; the seeded SSND2Net[P] nets back off for 10-25 steps before their first applied one
        sc = getattr(tr, "grad_scaler", None)
        scale0 = float(sc.get_scale()) if sc is not None else None
        t0 = time.perf_counter()
        for _ in range(a.steps):
            losses.append(float(tr.train_step(b)["loss"]))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        scale1 = float(sc.get_scale()) if sc is not None else None
        skipped = None
        if scale0 and scale1:
            import math
            skipped = max(0, int(round(math.log2(scale0 / scale1))))
        print(json.dumps({"model": name, "patch": a.size, "batch": a.batch, "ms_per_step": round(dt * 1e3, 2),
                          "patches_per_s": round(a.batch / dt, 3), "losses": [round(x, 4) for x in losses],
                          "hip_graph": bool(a.graph), "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                          "loss_scale": [scale0, scale1], "skipped_steps_in_timed_window": skipped}), flush=True)
        del tr
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
