"""Diagnostic (GPU box): after one hipGraph step + optimizer update, are the parameters or the replay broken?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training import zoo_trainers as Z

name = sys.argv[1]
plans, cfg, dj = nnunet_plans(2, (512, 512), batch_size=2)
torch.manual_seed(0)
tr = getattr(Z, "nnUNetTrainer" + name)(plans, cfg, 0, dj, device=torch.device("cuda"))
tr.initialize()
b = synthetic_batch(2, (512, 512), tr._get_deep_supervision_scales(), seed=3)
b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
for it in range(7):
    print("eager", it, float(tr.train_step(b)["loss"]), flush=True)
named = dict(tr.network.named_parameters())
before = {n: p.detach().clone() for n, p in named.items()}
tr.use_hip_graph = True
print("graph step", float(tr.train_step(b)["loss"]), "scale", tr.grad_scaler.get_scale(), flush=True)
worst = []
for n, p in named.items():
    d = (p.detach() - before[n]).abs().max().item()
    fin = bool(torch.isfinite(p).all())
    gfin = p.grad is None or bool(torch.isfinite(p.grad).all())
    worst.append((d if fin else float("inf"), n, fin, gfin, None if p.grad is None else p.grad.abs().max().item()))
worst.sort(key=lambda t: -t[0])
print("largest param changes:", worst[:8], flush=True)
print("nonfinite params:", sum(1 for w in worst if not w[2]), "nonfinite grads:", sum(1 for w in worst if not w[3]))
with torch.no_grad(), torch.autocast("cuda"):
    out = tr.network(b["data"])
    l = tr.loss(list(out), b["target"])
print("eager forward loss with current params", float(l), flush=True)
print("replay loss", float(tr._graphed(b["data"], b["target"]).detach().cpu()), flush=True)
bufs = {n: bool(torch.isfinite(v).all()) for n, v in tr.network.named_buffers()}
print("nonfinite buffers:", [n for n, ok in bufs.items() if not ok][:10])
