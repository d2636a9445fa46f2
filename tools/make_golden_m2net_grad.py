"""tests/golden/netgrad_M2Net_64.npz: whole-net BACKWARD of the reference's own M2Net (the benchmark model of BASELINE configs[2];
tools/make_golden.py gen_nets wrote this fixture for the small variant only) - eval mode, parameters by det_fill, the input of
net_M2Net_64.npz, loss = sum_i <out_i, G_i> / pixels with the formula-made G_i of gen_nets; dx in full, of every parameter
gradient <= 64 evenly strided samples and its L2 norm.  Run in the build container only:  python tools/make_golden_m2net_grad.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from make_golden import OUT, det_fill  # noqa: E402

if __name__ == "__main__":
    ref_shim.install()
    from nnunetv2.nets import m2net
    torch.manual_seed(0)
    net = m2net.M2Net(1, 2, True)
    det_fill(net)
    net.eval()
    x = torch.tensor(np.load(os.path.join(OUT, "net_M2Net_64.npz"))["x"]).requires_grad_(True)
    loss = 0
    for i, o in enumerate(net(x)):
        j = torch.arange(o.numel(), dtype=torch.float64)
        loss = loss + (o * torch.sin(0.37 * j + i).float().view_as(o)).sum() / o[0, 0].numel()
    loss.backward()
    gd, names = {"dx": x.grad.numpy()}, []
    for k, (n, p) in enumerate(net.named_parameters()):
        if p.grad is None:
            continue
        g = p.grad.reshape(-1)
        names.append(n)
        gd[f"g{k}"] = g[::max(1, g.numel() // 64)][:64].numpy()
        gd[f"n{k}"] = np.array(float(g.double().norm()))
    np.savez_compressed(os.path.join(OUT, "netgrad_M2Net_64.npz"), names=np.array(names), samples=np.array(64), **gd)
    print(len(names), "parameter gradients;", os.path.getsize(os.path.join(OUT, "netgrad_M2Net_64.npz")), "bytes")
