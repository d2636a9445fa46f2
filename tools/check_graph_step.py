"""Diagnostic (GPU box): gradients of the hipGraph-replayed forward+loss+backward (training/graph_step.py) against the
eager step on the same batch, train mode with DropPath disabled.  Usage: python tools/check_graph_step.py 512 M2NetP"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training import zoo_trainers as Z
from nnuzoo_amd.training.graph_step import GraphedForwardBackward

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
name = sys.argv[2] if len(sys.argv) > 2 else "M2NetP"
plans, cfg, dj = nnunet_plans(2, (size, size), batch_size=2)
torch.manual_seed(0)
tr = getattr(Z, "nnUNetTrainer" + name)(plans, cfg, 0, dj, device=torch.device("cuda"))
tr.initialize()
tr.network.train()
for m in tr.network.modules():
    if hasattr(m, "drop_prob"):
        m.drop_prob = 0.0
AC = tr.grad_scaler is not None
b = synthetic_batch(2, (size, size), tr._get_deep_supervision_scales(), seed=3)
data, target = b["data"].cuda(), [t.cuda() for t in b["target"]]
for p in tr.network.parameters():
    p.grad = None
with torch.autocast("cuda", enabled=AC):
    out = tr.network(data)
    l = tr.loss(list(out), target)
l.backward()
eager = {n: p.grad.clone() for n, p in tr.network.named_parameters() if p.grad is not None}
print("eager loss", l.item(), flush=True)
# yardstick: a SECOND eager pass on the same batch (fp32 atomics / reduction order make two runs differ)
for p in tr.network.parameters():
    p.grad = None
with torch.autocast("cuda", enabled=AC):
    out = tr.network(data)
    l = tr.loss(list(out), target)
l.backward()
nb, worst = 0, 0.0
gn = sum((eager[n].float() ** 2).sum() for n in eager).sqrt().item()
dn = 0.0
for n, p in tr.network.named_parameters():
    if p.grad is None or n not in eager:
        continue
    d = (p.grad - eager[n]).abs().max().item()
    s_ = eager[n].abs().max().item()
    dn += ((p.grad - eager[n]).float() ** 2).sum().item()
    if not (d <= 3e-2 * max(s_, 1e-6)):
        nb += 1
        worst = max(worst, d / max(s_, 1e-12))
print("eager vs eager: params off by > 3e-2:", nb, "worst", worst, "global rel L2", dn ** 0.5 / gn, flush=True)
del l, out   # the eager autograd graph (AccumulateGrad nodes bound to the default stream) must not outlive into the capture
for p in tr.network.parameters():
    p.grad = None
g = GraphedForwardBackward(tr.network, tr.loss, None, autocast=AC)
for it in range(3):
    lg = g(data, target)
    torch.cuda.synchronize()
    bad = []
    for n, p in tr.network.named_parameters():
        if p.grad is None or n not in eager:
            continue
        d = (p.grad - eager[n]).abs().max().item()
        s = eager[n].abs().max().item()
        if not (d <= 3e-2 * max(s, 1e-6)):
            bad.append((d / max(s, 1e-12), n))
    bad.sort(reverse=True)
    dn = sum(((p.grad - eager[n]).float() ** 2).sum().item() for n, p in tr.network.named_parameters()
             if p.grad is not None and n in eager)
    print("replay", it, "global rel L2 vs eager", dn ** 0.5 / gn, "loss", lg.item(), "memset nodes rewritten", g.memset_nodes_replaced,
          "params off by > 3e-2:", len(bad), bad[:5], flush=True)
    junk = [torch.full((1 + 37 * i,), float("nan"), device="cuda") for i in range(1500)]   # allocations between replays
    torch.cuda.synchronize()
    del junk
