"""The CPU oracle of SwT2Net (oracle/swt2net.py) against the REFERENCE's own outputs and autograd: the whole-net fixtures that
tools/make_golden.py wrote from /root/reference/nnunetv2/nets/swt2net.py (eval mode, parameters by golden_util.det_fill) - the
same fixtures the HIP path is held to in tests/test_zoo_gpu.py.  CPU only."""
import json
import os

import numpy as np
import torch

from golden_util import det_fill

G = os.path.join(os.path.dirname(__file__), "golden")


def _build():
    from oracle.swt2net import SwT2Net
    torch.manual_seed(0)
    net = SwT2Net(1, 2, True)
    det_fill(net)
    return net.eval()


def test_state_dict_keys_and_shapes_equal_the_references():
    man = json.load(open(os.path.join(G, "state_dict_manifest.json")))["SwT2Net"]
    assert [[k, list(v.shape)] for k, v in _build().state_dict().items()] == man


def test_forward_equals_the_references_outputs():
    z = np.load(os.path.join(G, "net_SwT2Net_64.npz"))
    with torch.no_grad():
        outs = _build()(torch.tensor(z["x"]))
    assert len(outs) == 7
    for i, o in enumerate(outs):
        ref = torch.tensor(z[f"out{i}"])
        err = (o - ref).abs().max().item()
        assert err <= 2e-4 * ref.abs().max().item() + 1e-5, (i, err, ref.abs().max().item())


def test_backward_equals_the_references_autograd():
    z = np.load(os.path.join(G, "netgrad_SwT2Net_64.npz"))
    x = torch.tensor(np.load(os.path.join(G, "net_SwT2Net_64.npz"))["x"]).requires_grad_(True)
    net = _build()
    loss = 0
    for i, o in enumerate(net(x)):
        j = torch.arange(o.numel(), dtype=torch.float64)
        loss = loss + (o * torch.sin(0.37 * j + i).float().view_as(o)).sum() / o[0, 0].numel()
    loss.backward()
    ref = torch.tensor(z["dx"])
    assert (x.grad - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
    assert [n for n, p in net.named_parameters() if p.grad is not None] == [str(n) for n in z["names"]]
    floor = 1e-8 * max(float(z[f"n{k}"]) for k, (n, p) in enumerate(net.named_parameters()) if p.grad is not None)
    for k, (n, p) in enumerate(net.named_parameters()):
        if p.grad is None:
            continue
        g = p.grad.reshape(-1)
        norm = float(z[f"n{k}"])
        assert abs(float(g.double().norm()) - norm) <= 4e-3 * norm + floor, (n, float(g.double().norm()), norm)
        samp = g[::max(1, g.numel() // 256)][:256]
        rs = torch.tensor(z[f"g{k}"])
        assert (samp - rs).abs().max().item() <= 4e-3 * max(rs.abs().max().item(), norm / max(1.0, g.numel() ** 0.5)) + floor, n
