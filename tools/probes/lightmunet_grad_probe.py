"""Which gradients of the stand-alone LightMUNet disagree with the reference fixture, and by how much?
Usage (GPU box): python tools/probes/lightmunet_grad_probe.py [2d|3d]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import det_fill
from test_lightmunet import _build

tag = sys.argv[1] if len(sys.argv) > 1 else "2d"
z = np.load(os.path.join(ROOT, "tests", "golden", f"net_LightMUNet_{tag}.npz"))
net = _build(tag)
det_fill(net)
with torch.no_grad():
    for n, p in net.named_parameters():
        if n.endswith("A_log"):
            p.copy_(torch.log(1.0 + torch.arange(p.numel(), dtype=torch.float32).reshape(p.shape) % 16) * 0.9 + 0.05 * p)
net = net.cuda().train()
x = torch.tensor(z["x"]).cuda().requires_grad_(True)
y = net(x)
print("forward max err / rms", (y.detach().cpu() - torch.tensor(z["y"])).abs().max().item() / torch.tensor(z["y"]).pow(2).mean().sqrt().item())
j = torch.arange(y.numel(), dtype=torch.float64)
((y * torch.sin(0.37 * j).float().view_as(y).cuda()).sum() / y[0, 0].numel()).backward()
rdx = torch.tensor(z["dx"])
print("dx: max err", (x.grad.cpu() - rdx).abs().max().item(), "max ref", rdx.abs().max().item(), "norm got/ref",
      x.grad.norm().item(), rdx.norm().item())
rows = []
for k, (n, p) in enumerate(net.named_parameters()):
    if p.grad is None:
        continue
    want = float(z[f"n{k}"])
    got = p.grad.double().norm().item()
    rows.append((abs(got - want) / (want + 1e-30), n, got, want))
rows.sort(reverse=True)
for r in rows[:25]:
    print(f"{r[0]:9.2e}  {r[1]:60s} got {r[2]:.4e} ref {r[3]:.4e}")
print("median rel err", sorted(r[0] for r in rows)[len(rows) // 2])
