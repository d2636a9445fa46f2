"""Which parameter gradients of the whole-net backward fixture (tests/golden/netgrad_<name>_64.npz, the reference's own autograd) the
HIP path misses by how much, twice in a row (run-to-run spread beside the error).  Usage (GPU box): python tools/probes/m2net_grad_probe.py [M2Net]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from golden_util import det_fill  # noqa: E402
from nnuzoo_amd.nets import m2net  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "M2Net"
G = os.path.join(ROOT, "tests", "golden")
z = np.load(os.path.join(G, f"netgrad_{name}_64.npz"))
nsamp = int(z["samples"]) if "samples" in z else 256
x0 = np.load(os.path.join(G, f"net_{name}_64.npz"))["x"]


def run():
    torch.manual_seed(0)
    net = getattr(m2net, name)(1, 2, True)
    det_fill(net)
    net = net.cuda().eval()
    x = torch.tensor(x0).cuda().requires_grad_(True)
    outs = net(x)
    loss = 0
    for i, o in enumerate(outs):
        j = torch.arange(o.numel(), dtype=torch.float64)
        loss = loss + (o * torch.sin(0.37 * j + i).float().view_as(o).cuda()).sum() / o[0, 0].numel()
    loss.backward()
    return net, x.grad.detach().cpu()


nets = [run(), run()]
top = max(float(z[f"n{k}"]) for k, (n, p) in enumerate(nets[0][0].named_parameters()) if p.grad is not None)
rows = []
for k, ((n, p), (_, q)) in enumerate(zip(nets[0][0].named_parameters(), nets[1][0].named_parameters())):
    if p.grad is None:
        continue
    g, g2 = p.grad.reshape(-1), q.grad.reshape(-1)
    ref = torch.tensor(z[f"g{k}"])
    st = max(1, g.numel() // nsamp)
    got, got2 = g[::st][:nsamp].float().cpu(), g2[::st][:nsamp].float().cpu()
    norm = float(z[f"n{k}"])
    scale = max(norm / g.numel() ** 0.5, ref.abs().max().item(), 1e-8 * top)
    rows.append(((got - ref).abs().max().item() / scale, (got - got2).abs().max().item() / scale, norm / top, abs(float(g.double().norm()) - norm) / (norm + 1e-30), n, tuple(p.shape)))
rows.sort(reverse=True)
print("dx err", ((nets[0][1] - torch.tensor(z["dx"])).abs().max() / torch.tensor(z["dx"]).abs().max()).item())
print(f"{'err':>9s} {'run2run':>9s} {'norm/top':>9s} {'normerr':>9s}  parameter")
for r in rows[:40]:
    print(f"{r[0]:9.2e} {r[1]:9.2e} {r[2]:9.2e} {r[3]:9.2e}  {r[4]} {r[5]}")
print("parameters above 1e-2:", sum(r[0] > 1e-2 for r in rows), "of", len(rows))
