"""The CPU oracle of the zoo's flagship net (oracle/m2net.py) against the REFERENCE's own outputs: the whole-net fixtures that
tools/make_golden.py wrote from /root/reference/nnunetv2/nets/m2net.py (eval mode, parameters filled by golden_util.det_fill) -
the same fixtures the HIP path is held to in tests/test_zoo_gpu.py.  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import det_fill

G = os.path.join(os.path.dirname(__file__), "golden")


def _build(name):
    from oracle import m2net
    torch.manual_seed(0)
    net = {"M2NetP": m2net.M2NetP, "M2Net": m2net.M2Net}[name](1, 2, True)
    det_fill(net)
    return net.eval()


@pytest.mark.parametrize("name", ["M2NetP", "M2Net"])
def test_state_dict_keys_and_shapes_equal_the_references(name):
    man = json.load(open(os.path.join(G, "state_dict_manifest.json")))[name]
    sd = _build(name).state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == man


@pytest.mark.parametrize("name", ["M2NetP", "M2Net"])
def test_forward_equals_the_references_outputs(name):
    z = np.load(os.path.join(G, f"net_{name}_64.npz"))
    net = _build(name)
    with torch.no_grad():
        outs = net(torch.tensor(z["x"]))
    assert len(outs) == 7
    for i, o in enumerate(outs):
        ref = torch.tensor(z[f"out{i}"])
        err = (o - ref).abs().max().item()
        assert err <= 2e-4 * ref.abs().max().item() + 1e-5, (name, i, err, ref.abs().max().item())


def test_backward_equals_the_references_autograd():
    """dx in full; of every parameter gradient the reference's <= 256 strided samples and its L2 norm (netgrad_M2NetP_64.npz)"""
    z = np.load(os.path.join(G, "netgrad_M2NetP_64.npz"))
    x = torch.tensor(np.load(os.path.join(G, "net_M2NetP_64.npz"))["x"]).requires_grad_(True)
    net = _build("M2NetP")
    loss = 0
    for i, o in enumerate(net(x)):
        j = torch.arange(o.numel(), dtype=torch.float64)
        loss = loss + (o * torch.sin(0.37 * j + i).float().view_as(o)).sum() / o[0, 0].numel()
    loss.backward()
    ref = torch.tensor(z["dx"])
    assert (x.grad - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
    names = [str(n) for n in z["names"]]
    assert [n for n, p in net.named_parameters() if p.grad is not None] == names
    for k, (n, p) in enumerate(net.named_parameters()):
        if p.grad is None:
            continue
        g = p.grad.reshape(-1)
        norm = float(z[f"n{k}"])
        assert abs(float(g.double().norm()) - norm) <= 2e-3 * norm + 1e-7, (n, float(g.double().norm()), norm)
        samp = g[::max(1, g.numel() // 256)][:256]
        rs = torch.tensor(z[f"g{k}"])
        assert (samp - rs).abs().max().item() <= 2e-3 * max(rs.abs().max().item(), norm / max(1.0, g.numel() ** 0.5)) + 1e-7, n


def test_bench_secondary_cpu_baseline_full_step():
    """bench.py's `secondary.cpu_baseline`: one full oracle step at 64^2 (budget 0: the 128^2 step is left out here)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    import bench
    r = bench.cpu_m2net_step_baseline(budget_s=0.0)
    assert r["kind"] == "port" and r["unit"] == "patches/s" and r["cores"] >= 1
    assert list(r["step_seconds"]) == ["64"] and 0 < r["value"] < 1.0
