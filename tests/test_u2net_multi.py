"""N-D U^2-Net / U^2-Net-P on monai's Convolution unit (nnuzoo_amd/nets/u2net_multi.py, trainers nnUNetTrainerU2NetMulti[P]) against
fixtures produced by the REFERENCE's own classes (tools/make_golden_u2net_multi.py: /root/reference/nnunetv2/nets/u2net_multi.py
U2NET / U2NETP imported under tools/ref_shim.py, monai's `Convolution` served by the restatement in that script - monai is absent,
SURVEY 8c: the fixtures pin the reference's wiring and registration order, not monai's internals):
  CPU  state_dict names / shapes / ORDER, torch.manual_seed(0) + class + He init = the reference's parameters bit for bit; forward
       (seven outputs) + backward (dx, every parameter gradient: L2 norm + 256 strided samples) in fp32, 2-D (both nets) and 3-D (P)
  GPU  the same in fp32 on the device; under fp16 autocast the conv -> BatchNorm -> ReLU units of U2NET with channel counts in
       multiples of 32 run on the tap-table MFMA conv kernels (backend asserted); trainer steps; nnUNetTrainerSwUNETR fails the way
       the reference does without monai."""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import det_fill

G = os.path.join(os.path.dirname(__file__), "golden")
MAN = json.load(open(os.path.join(G, "u2net_multi_manifest.json")))
CASES = [("U2NETmulti", 2), ("U2NETPmulti", 2), ("U2NETPmulti", 3)]


def _build(name, nd):
    from nnuzoo_amd.nets import u2net_multi as um
    from nnuzoo_amd.utilities.network_initialization import InitWeights_He
    torch.manual_seed(0)
    net = (um.U2NET if name == "U2NETmulti" else um.U2NETP)(spatial_dims=nd, in_ch=1, out_ch=2, deep_supervision=True)
    net.apply(InitWeights_He(1e-2))
    return net


@pytest.mark.parametrize("name,nd", CASES)
def test_state_dict_and_seeded_construction(name, nd):
    from test_u2net_swt import _digest
    net = _build(name, nd)
    want = MAN[f"{name}_{nd}d"]
    assert [[k, list(v.shape)] for k, v in net.state_dict().items()] == want["state_dict"]
    got = _digest(net.state_dict())
    assert got["n_tensors"] == want["seeded"]["n_tensors"] and got["sha256"] == want["seeded"]["sha256"]


def _fwd_bwd(name, nd, dev, rtol_out, rtol_grad):
    z = np.load(os.path.join(G, f"net_{name}_{nd}d.npz"))
    net = _build(name, nd)
    det_fill(net)
    net = net.to(dev).eval()
    x = torch.tensor(z["x"]).to(dev).requires_grad_(True)
    outs = list(net(x))
    assert len(outs) == 7
    loss = 0
    for i, o in enumerate(outs):
        assert list(o.shape) == list(z[f"shape{i}"])
        ref = torch.tensor(z[f"out{i}"])
        got = o.detach().float().cpu().reshape(-1)[::int(z[f"stride{i}"])]
        err = (got - ref).abs().max().item()
        assert err <= rtol_out * max(ref.abs().max().item(), 1e-6), (name, nd, i, err)
        j = torch.arange(o.numel(), dtype=torch.float64)
        loss = loss + (o.float() * torch.sin(0.37 * j + i).float().view_as(o).to(dev)).sum() / o[0, 0].numel()
    loss.backward()
    dref = torch.tensor(z["dx"])
    derr = (x.grad.float().cpu() - dref).abs().max().item()
    # (dx passes through the first unit's mean-removing norm over the whole map: at 64^3 two CPU runs with different thread counts
    #  already differ by 8e-4 of its range)
    assert derr <= 10 * rtol_grad * dref.abs().max().item(), (name, nd, "dx", derr)
    names = [str(n) for n in z["names"]]
    assert [n for n, p in net.named_parameters() if p.grad is not None] == names
    top = max(float(z[f"n{k}"]) for k, (n, p) in enumerate(net.named_parameters()) if p.grad is not None)
    for k, (n, p) in enumerate(net.named_parameters()):
        if p.grad is None:
            continue
        want = float(z[f"n{k}"])
        have = p.grad.double().norm().item()
        # (a conv bias in front of a mean-removing norm has a gradient that is pure cancellation: floor at 1e-4 of the largest norm)
        assert abs(have - want) <= 5 * rtol_grad * max(want, 1e-4 * top), (name, nd, n, have, want)


@pytest.mark.parametrize("name,nd", CASES)
def test_cpu_forward_backward_golden(name, nd):
    # 3-D: the PReLU slope / norm gradients are fp32 sums over 64^3 x 64 values; two CPU runs with different thread counts differ by
    # 1.2e-3 there (the generator ran on 2 threads)
    _fwd_bwd(name, nd, "cpu", 2e-5, 2e-4 if nd == 2 else 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("name,nd", [c for c in CASES if c[1] == 2])
def test_gpu_forward_backward_golden(hip_lib, name, nd):
    # fp32 on the device = the library's convolutions (the native unit is the fp16 autocast step below); same tolerances as
    # tests/test_u2net_swt.py holds nets/u2net.py to.  (The 3-D net is stock torch on the device - no kernel of this package - and the
    # library's solver search for its 64^3 convolutions costs a fresh box a minute: its wiring is pinned by the CPU test above.)
    _fwd_bwd(name, nd, "cuda", 3e-4, 2e-2)


@pytest.mark.gpu
def test_autocast_batchnorm_units_run_on_hip(hip_lib):
    """eval-mode forward under fp16 autocast against the fp32 golden at fp16 tolerance; the conv -> BatchNorm -> ReLU units whose
    channel counts are multiples of 32 (RSU6 ... RSU4F of U2NET) report the HIP backend, RSU7's InstanceNorm + PReLU units the library"""
    from nnuzoo_amd.nets.u2net_multi import Convolution
    z = np.load(os.path.join(G, "net_U2NETmulti_2d.npz"))
    net = _build("U2NETmulti", 2)
    det_fill(net)
    net = net.cuda().eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        outs = net(torch.tensor(z["x"]).cuda())
    for i, o in enumerate(outs):
        ref = torch.tensor(z[f"out{i}"])
        got = o.float().cpu().reshape(-1)[::int(z[f"stride{i}"])]
        assert (got - ref).abs().max().item() <= 3e-2 * ref.abs().max().item(), i
    units = [m for m in net.modules() if isinstance(m, Convolution) and "adn" in m._modules]
    assert sum(m.backend == "hip" for m in units) >= 40, sum(m.backend == "hip" for m in units)
    assert net.stage1.rebnconv1.backend == "library"          # RSU7: InstanceNorm + PReLU
    assert net.stage2.rebnconv2.backend == "hip"              # RSU6: 32 -> 32 conv + BatchNorm + ReLU


@pytest.mark.gpu
@pytest.mark.parametrize("trainer,nd", [("nnUNetTrainerU2NetMulti", 2), ("nnUNetTrainerU2NetMultiP", 2)])
def test_trainer_steps(hip_lib, trainer, nd):
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training import zoo_trainers as Z
    size = (64,) * nd
    plans, cfg, dj = nnunet_plans(nd, size, batch_size=2)
    torch.manual_seed(0)
    tr = getattr(Z, trainer)(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    scales = tr._get_deep_supervision_scales()
    assert scales == [[1.0] * nd] * 7
    b = synthetic_batch(2, size, scales, seed=5)
    b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
    before = [p.detach().clone() for p in tr.network.parameters()]
    losses = [float(tr.train_step(b)["loss"]) for _ in range(16)]
    assert all(np.isfinite(losses)), losses
    # (a conv bias in front of BatchNorm has an identically zero gradient - mean removal - and He init leaves it at zero: AdamW's
    #  decay of zero is zero, so those tensors legitimately stay put; everything else must move at lr 1e-4)
    pairs = [(a, p.detach()) for a, p in zip(before, tr.network.parameters()) if bool(a.any()) or bool(p.detach().any())]
    moved = sum(int(not torch.equal(a, p)) for a, p in pairs)
    assert len(pairs) > 0.75 * len(before) and moved > 0.9 * len(pairs), (moved, len(pairs), len(before), losses,
                                                                          tr.grad_scaler.get_scale())


def test_swunetr_plugin_fails_like_the_reference_without_monai():
    """nnUNetTrainerSwUNETR.py:4 imports monai's SwinUNETR at module level; here the plugin class exists under its name with the
    reference's hyper-parameters and raises the same exception type where the network would be built"""
    from nnuzoo_amd.synthetic import nnunet_plans
    from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerSwUNETR
    plans, cfg, dj = nnunet_plans(2, (64, 64), batch_size=2)
    tr = nnUNetTrainerSwUNETR(plans, cfg, 0, dj, device=torch.device("cpu"))
    assert tr.initial_lr == 1e-4 and tr.weight_decay == 5e-2 and tr.enable_deep_supervision is False
    assert tr._get_deep_supervision_scales() is None
    try:
        import monai  # noqa: F401
    except ImportError:
        with pytest.raises(ModuleNotFoundError, match="monai"):
            tr.initialize()
