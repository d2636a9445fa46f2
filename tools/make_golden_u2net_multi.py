"""Round-6 fixtures for nets/u2net_multi.py (U2NET / U2NETP on monai's Convolution unit, 2-D and 3-D), generated in the BUILD
CONTAINER from the reference's own module imported under tools/ref_shim.py; only arrays / shapes / digests are stored.
    python tools/make_golden_u2net_multi.py
monai is absent: the reference's `Convolution(...)` calls are served by the restatement below (the published monai 1.3
block: conv with same padding + ADN in "NDA" order; Norm.INSTANCE = InstanceNorm without affine, Norm.BATCH = BatchNorm, Act.PRELU
= nn.PReLU() with one parameter at 0.25, "relu" = nn.ReLU) - so the fixtures pin the REFERENCE's wiring (unit / stage order, channel
plan, the act / norm arguments RSU7 drops, pooling, up-sampling, the side units of U2NETP) and the parameter registration order,
not monai's internals.
  * net_U2NETmulti_{2d}.npz, net_U2NETPmulti_{2d,3d}.npz   x, the seven outputs (eval mode, golden_util.det_fill parameters), dx,
                                                           L2 norm + 256 strided samples of every parameter gradient
  * u2net_multi_manifest.json                              state_dict names / shapes / order + sha256 of the state_dict after
                                                           torch.manual_seed(0) + the reference's class and He init"""
import json
import os
import sys

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_shim  # noqa: E402
from golden_util import det_fill  # noqa: E402
from make_golden import state_digest  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class ADN(nn.Sequential):
    def __init__(self, nd, channels, act, norm):
        super().__init__()
        n = (norm.value if hasattr(norm, "value") else norm).upper()
        a = (act.value if hasattr(act, "value") else act).upper()
        self.add_module("N", {("INSTANCE", 2): nn.InstanceNorm2d, ("INSTANCE", 3): nn.InstanceNorm3d,
                              ("BATCH", 2): nn.BatchNorm2d, ("BATCH", 3): nn.BatchNorm3d}[(n, nd)](channels))
        self.add_module("A", {"PRELU": nn.PReLU, "RELU": nn.ReLU}[a]())


class Convolution(nn.Sequential):
    """monai.networks.blocks.convolutions.Convolution, the arguments u2net_multi.py uses"""

    def __init__(self, spatial_dims, in_channels, out_channels, strides=1, kernel_size=3, adn_ordering="NDA", act="PRELU",
                 norm="INSTANCE", dropout=None, dropout_dim=1, dilation=1, groups=1, bias=True, conv_only=False,
                 is_transposed=False, padding=None, output_padding=None):
        super().__init__()
        assert adn_ordering == "NDA" and dropout is None and not is_transposed and groups == 1
        pad = (kernel_size - 1) // 2 * dilation if padding is None else padding
        conv = {2: nn.Conv2d, 3: nn.Conv3d}[spatial_dims]
        self.add_module("conv", conv(in_channels, out_channels, kernel_size, strides, pad, dilation, groups, bias))
        if not conv_only:
            self.add_module("adn", ADN(spatial_dims, out_channels, act, norm))


def main():
    ref_shim.install()
    import monai.networks.blocks.convolutions as mc
    mc.Convolution = Convolution
    from nnunetv2.nets import u2net_multi as ref
    from nnunetv2.utilities.network_initialization import InitWeights_He
    man = {}
    for name, cls, nd, size in [("U2NETmulti", ref.U2NET, 2, 64), ("U2NETPmulti", ref.U2NETP, 2, 64),
                                ("U2NETPmulti", ref.U2NETP, 3, 64)]:
        torch.manual_seed(0)
        net = cls(spatial_dims=nd, in_ch=1, out_ch=2, deep_supervision=True)
        net.apply(InitWeights_He(1e-2))
        man[f"{name}_{nd}d"] = {"state_dict": [[k, list(v.shape)] for k, v in net.state_dict().items()],
                                "seeded": state_digest(net.state_dict())}
        det_fill(net)
        net.eval()
        x = torch.randn(1, 1, *([size] * nd), generator=torch.Generator().manual_seed(31 + nd))
        xg = x.clone().requires_grad_(True)
        outs = list(net(xg))
        stride = [max(1, o.numel() // 16384) for o in outs]
        loss = 0
        for i, o in enumerate(outs):
            j = torch.arange(o.numel(), dtype=torch.float64)
            loss = loss + (o * torch.sin(0.37 * j + i).float().view_as(o)).sum() / o[0, 0].numel()
        loss.backward()
        gd = {"x": x.numpy(), "dx": xg.grad.numpy()}
        for i, o in enumerate(outs):
            gd[f"out{i}"] = o.detach().reshape(-1)[::stride[i]].numpy()
            gd[f"shape{i}"] = np.array(o.shape)
            gd[f"stride{i}"] = np.array(stride[i])
        names = []
        for k, (n, p) in enumerate(net.named_parameters()):
            if p.grad is None:
                continue
            g = p.grad.reshape(-1)
            names.append(n)
            gd[f"g{k}"] = g[::max(1, g.numel() // 256)][:256].numpy()
            gd[f"n{k}"] = np.array(float(g.double().norm()))
        np.savez_compressed(os.path.join(OUT, f"net_{name}_{nd}d.npz"), names=np.array(names), **gd)
        print(name, nd, "params", sum(p.numel() for p in net.parameters()), "outs", [tuple(o.shape) for o in outs], flush=True)
    with open(os.path.join(OUT, "u2net_multi_manifest.json"), "w") as f:
        json.dump(man, f)


if __name__ == "__main__":
    main()
