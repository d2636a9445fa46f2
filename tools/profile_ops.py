"""Host-side op census of one zoo training step (torch.profiler): which ATen ops are launched how often, and which
python call sites issue the zero fills / copies.  Usage (GPU box): python tools/profile_ops.py [--model M2Net]"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training import zoo_trainers as Z


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="M2Net")
    ap.add_argument("--size", type=int, default=512)
    a = ap.parse_args()
    cls = getattr(Z, "nnUNetTrainer" + a.model)
    plans, cfg, dj = nnunet_plans(2, (a.size, a.size), batch_size=2)
    torch.manual_seed(0)
    tr = cls(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    tr.use_hip_graph = False   # the census is about the eager op stream
    b = synthetic_batch(2, (a.size, a.size), tr._get_deep_supervision_scales(), seed=3)
    b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
    for _ in range(3):
        tr.train_step(b)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        tr.train_step(b)
        torch.cuda.synchronize()
    ev = prof.events()
    cnt = collections.Counter(e.name for e in ev)
    print("== top ops by count")
    for n, c in cnt.most_common(45):
        print(f"{c:6d}  {n}")
    sites = collections.Counter()
    for e in ev:
        if e.name in ("aten::zeros", "aten::zero_", "aten::zeros_like", "aten::new_zeros", "aten::fill_"):
            st = [s for s in (e.stack or []) if "nnuzoo_amd" in s or "torch/optim" in s or "torch/amp" in s or "clip_grad" in s]
            sites[(e.name, st[0] if st else "(autograd / internal)")] += 1
    print("== top ops by self CPU time")
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=60))
    print("== zero-fill call sites")
    for (n, s), c in sites.most_common(40):
        print(f"{c:6d}  {n:18s} {s}")


if __name__ == "__main__":
    main()
