"""Where does the SwT2Net backward at 2 x 512^2 differ from the CPU oracle (round 6)?  dx and per-parameter gradient norms in eval and
training mode, at 512^2 and 128^2.  Usage (GPU box): python tools/probes/swt2net_fullshape_bwd_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from golden_util import det_fill
from oracle.swt2net import SwT2Net as Ref
from nnuzoo_amd.nets.swt2net import SwT2Net
from nnuzoo_amd.synthetic import synthetic_batch
from nnuzoo_amd.token_linear import deferred_wgrads


def functional(outs, dev):
    tot = 0
    for i, o in enumerate(outs):
        j = torch.arange(o.numel(), dtype=torch.float64)
        tot = tot + (o.float() * torch.sin(0.37 * j + i).float().view_as(o).to(dev)).sum() / o[0, 0].numel()
    return tot


def run(size, train):
    torch.manual_seed(0)
    ref = Ref(1, 2, True)
    det_fill(ref)
    net = SwT2Net(1, 2, True)
    net.load_state_dict(ref.state_dict())
    for m in list(ref.modules()) + list(net.modules()):
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
    ref.train(train)
    net = net.cuda().train(train)
    x = synthetic_batch(2, (size, size), [[1, 1]], seed=11)["data"]
    xr = x.clone().requires_grad_(True)
    functional(ref(xr), "cpu").backward()
    # the oracle's own backward conditioning: dx at a 1e-6 perturbed input
    xp = (x * (1 + 1e-6)).requires_grad_(True)
    ref.zero_grad()
    functional(ref(xp), "cpu").backward()
    bsens = (xp.grad - xr.grad).abs().max().item() / xr.grad.abs().max().item()
    ref.zero_grad()
    xr = x.clone().requires_grad_(True)
    functional(ref(xr), "cpu").backward()
    xd = x.cuda().requires_grad_(True)
    with deferred_wgrads():
        functional(net(xd), "cuda").backward()
    derr = (xd.grad.cpu() - xr.grad).abs().max().item() / xr.grad.abs().max().item()
    want = {n: p.grad.double().norm().item() for n, p in ref.named_parameters() if p.grad is not None}
    top = max(want.values())
    devs = sorted(((abs(p.grad.double().norm().item() - want[n]) / max(want[n], 1e-4 * top), n)
                   for n, p in net.named_parameters() if n in want), reverse=True)
    print(f"{size}^2 train={train}: dx err {derr:.2e} of range; the oracle's own dx response to a 1e-6 input change {bsens:.2e}; "
          f"worst gradient-norm deviations {[(round(d, 5), n) for d, n in devs[:4]]}", flush=True)


for size, train in ((128, False), (128, True), (512, False), (512, True)):
    run(size, train)
