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


@pytest.mark.parametrize("name", ["M2NetP", "M2Net"])
def test_backward_equals_the_references_autograd(name):
    """dx in full; of every parameter gradient the reference's strided samples and its L2 norm (netgrad_M2NetP_64.npz from
    tools/make_golden.py, netgrad_M2Net_64.npz - the benchmark model - from tools/make_golden_m2net_grad.py)"""
    z = np.load(os.path.join(G, f"netgrad_{name}_64.npz"))
    nsamp = int(z["samples"]) if "samples" in z else 256
    x = torch.tensor(np.load(os.path.join(G, f"net_{name}_64.npz"))["x"]).requires_grad_(True)
    net = _build(name)
    loss = 0
    for i, o in enumerate(net(x)):
        j = torch.arange(o.numel(), dtype=torch.float64)
        loss = loss + (o * torch.sin(0.37 * j + i).float().view_as(o)).sum() / o[0, 0].numel()
    loss.backward()
    ref = torch.tensor(z["dx"])
    assert (x.grad - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
    names = [str(n) for n in z["names"]]
    assert [n for n, p in net.named_parameters() if p.grad is not None] == names
    # tolerance: 4e-3 of each gradient's own norm (measured worst: 2.3e-3, hundreds of fp32 layers and the reference's scan sums
    # in another order) + 1e-8 of the LARGEST gradient norm of the net (gradients 8-11 orders below it are cancellation noise)
    floor = 1e-8 * max(float(z[f"n{k}"]) for k, (n, p) in enumerate(net.named_parameters()) if p.grad is not None)
    for k, (n, p) in enumerate(net.named_parameters()):
        if p.grad is None:
            continue
        g = p.grad.reshape(-1)
        norm = float(z[f"n{k}"])
        assert abs(float(g.double().norm()) - norm) <= 4e-3 * norm + floor, (n, float(g.double().norm()), norm)
        samp = g[::max(1, g.numel() // nsamp)][:nsamp]
        rs = torch.tensor(z[f"g{k}"])
        assert (samp - rs).abs().max().item() <= 4e-3 * max(rs.abs().max().item(), norm / max(1.0, g.numel() ** 0.5)) + floor, n


def test_bench_secondary_cpu_baseline_full_step():
    """bench.py's `secondary.cpu_baseline`: one full oracle step at 64^2 (budget 0: the 128^2 step is left out here)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    import bench
    r = bench.cpu_m2net_step_baseline(budget_s=0.0)
    assert r["kind"] == "port" and r["unit"] == "patches/s" and r["cores"] >= 1
    assert list(r["step_seconds"]) == ["64"] and 0 < r["value"] < 1.0


def test_training_trajectory_follows_the_references_own_cpu_run():
    """tests/golden/traj_m2netp_64.json: losses of the reference's own M2NetP + loss classes trained for 6 steps on the CPU
    (tools/dice_ref_cpu_zoo.py).  The oracle starts from the same seeded construction (the product's constructor draws the
    reference's RNG stream bit for bit - tests/golden/seeded_init.json - and its state_dict loads into the oracle), sees the same
    batches and takes the same AdamW steps.  Step 0 (same weights) agrees to fp32 rounding; after that two CPU fp32 runs of the
    SAME algorithm drift apart by 2e-3, 1e-2, 1e-2, 1e-2, 2e-2 (measured over all six steps) - AdamW's g / (sqrt(v) + eps) turns
    rounding-level gradient differences into O(lr) parameter differences - which is the size of the HIP path's own deviations
    from this fixture (tests/test_zoo_trajectory_gpu.py).  Three steps here (20 s each)."""
    from oracle import m2net as om
    from oracle.losses import deep_supervision_loss
    from nnuzoo_amd.nets.m2net import M2NetP
    from nnuzoo_amd.synthetic import synthetic_batch
    ref = json.load(open(os.path.join(G, "traj_m2netp_64.json")))
    torch.manual_seed(0)
    seeded = M2NetP(1, 2, True)
    net = om.M2NetP(1, 2, True)
    net.load_state_dict(seeded.state_dict())
    for m in net.modules():
        if type(m).__name__ == "StochasticDepth":
            m.p = 0.0
    net.train()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=5e-2, eps=1e-5, betas=(0.9, 0.999))
    got = []
    for it in range(3):
        b = synthetic_batch(2, (ref["size"], ref["size"]), ref["scales"], seed=1000 + it)
        opt.zero_grad(set_to_none=True)
        loss = deep_supervision_loss(list(net(b["data"])), b["target"], batch_dice=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
        opt.step()
        got.append(float(loss.detach()))
    want = ref["losses"][:3]
    assert abs(got[0] - want[0]) <= 2e-6, (got, want)
    assert abs(got[1] - want[1]) <= 1e-2 and abs(got[2] - want[2]) <= 3e-2, (got, want)
