"""Round-3 fixtures, generated in the BUILD CONTAINER from the reference's own modules (imported under tools/ref_shim.py);
only arrays / shapes / digests are stored (tests/golden/), no reference source travels.
    python tools/make_golden_r3.py
  * net_U2NET_64.npz / net_U2NETP_64.npz          forward of nets/u2net.py U2NET / U2NETP (eval, deterministic parameter
                                                  fill of tests/golden_util.py), 7 outputs each, + dx and the L2 norm /
                                                  256 strided samples of every parameter gradient (netgrad_*.npz)
  * net_SwinTransformerUnet_64.npz                forward of nets/swt.py get_swin_transformer_unet's network (eval)
  * r3_manifest.json                              state_dict names / shapes / ORDER of the three networks + sha256 digests
                                                  of their state_dict after torch.manual_seed(0) + the reference factories
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_shim  # noqa: E402
from golden_util import det_fill  # noqa: E402
from make_golden import state_digest  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    ref_shim.install()
    from nnunetv2.nets import swt, u2net
    man = {}
    for name, factory, size in [("U2NET", u2net.get_u2net_from_plans, 64), ("U2NETP", u2net.get_u2netp_from_plans, 64),
                                ("SwinTransformerUnet", swt.get_swin_transformer_unet, 64)]:
        torch.manual_seed(0)
        net = factory(2, 1, True, False)
        man[name] = {"state_dict": [[k, list(v.shape)] for k, v in net.state_dict().items()],
                     "seeded": state_digest(net.state_dict())}
        det_fill(net)
        net.eval()
        x = torch.randn(1, 1, size, size, generator=torch.Generator().manual_seed(31))
        xg = x.clone().requires_grad_(True)
        outs = net(xg)
        outs = outs if isinstance(outs, (tuple, list)) else [outs]
        np.savez_compressed(os.path.join(OUT, f"net_{name}_{size}.npz"), x=x.numpy(),
                            **{f"out{i}": o.detach().numpy() for i, o in enumerate(outs)})
        loss = 0
        for i, o in enumerate(outs):
            j = torch.arange(o.numel(), dtype=torch.float64)
            loss = loss + (o * torch.sin(0.37 * j + i).float().view_as(o)).sum() / o[0, 0].numel()
        loss.backward()
        gd = {"dx": xg.grad.numpy()}
        names = []
        for k, (n, p) in enumerate(net.named_parameters()):
            if p.grad is None:
                continue
            g = p.grad.reshape(-1)
            names.append(n)
            gd[f"g{k}"] = g[::max(1, g.numel() // 256)][:256].numpy()
            gd[f"n{k}"] = np.array(float(g.double().norm()))
        np.savez_compressed(os.path.join(OUT, f"netgrad_{name}_{size}.npz"), names=np.array(names), **gd)
        print(name, "params", sum(p.numel() for p in net.parameters()), "outs", [tuple(o.shape) for o in outs])
    with open(os.path.join(OUT, "r3_manifest.json"), "w") as f:
        json.dump(man, f)


if __name__ == "__main__":
    main()
