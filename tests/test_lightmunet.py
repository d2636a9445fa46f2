"""The stand-alone LightMUNet (nnuzoo_amd/nets/lightmunet.py, plugin nnUNetTrainerLightMUNet) against fixtures produced by the
REFERENCE's own class (tools/make_golden_lm2net.py lightmunet: nets/LightMUNet.py with mamba_ssm.Mamba bound to the
reference's vendored block on its selective_scan_ref), in the trainer's configuration (init_filters 32, blocks (1, 2, 2, 4)):
  CPU  state_dict names / shapes / ORDER, 2-D and 3-D
  GPU  forward (2e-4 of the rms), gradient structure and scale (the fixture's gradient values are ill-conditioned, see the
       test), 2-D 64^2 and 3-D 16^3; a trainer step each
monai's get_upsample_layer / get_norm_layer / get_act_layer are restated identically on both sides (unpinned)."""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import det_fill

G = os.path.join(os.path.dirname(__file__), "golden")
MAN = json.load(open(os.path.join(G, "lightmunet_manifest.json")))
CFG = {"2d": (2, 1), "3d": (3, 2)}


def _build(tag):
    from nnuzoo_amd.nets.lightmunet import LightMUNet
    sd, cin = CFG[tag]
    torch.manual_seed(0)
    return LightMUNet(spatial_dims=sd, init_filters=32, in_channels=cin, out_channels=3, blocks_down=[1, 2, 2, 4],
                      blocks_up=[1, 1, 1])


@pytest.mark.parametrize("tag", ["2d", "3d"])
def test_state_dict_manifest(tag):
    assert [[k, list(v.shape)] for k, v in _build(tag).state_dict().items()] == MAN[tag]


def test_mae_option_is_refused():
    from nnuzoo_amd.nets.lightmunet import LightMUNet
    with pytest.raises(NotImplementedError):
        LightMUNet(spatial_dims=2, mae=True)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["2d", "3d"])
def test_forward_backward_golden(hip_lib, tag):
    z = np.load(os.path.join(G, f"net_LightMUNet_{tag}.npz"))
    net = _build(tag)
    det_fill(net)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("A_log"):
                p.copy_(torch.log(1.0 + torch.arange(p.numel(), dtype=torch.float32).reshape(p.shape) % 16) * 0.9 + 0.05 * p)
    net = net.cuda().train()
    x = torch.tensor(z["x"]).cuda().requires_grad_(True)
    y = net(x)
    ref = torch.tensor(z["y"])
    assert y.shape == ref.shape
    rms = ref.pow(2).mean().sqrt().item()
    assert (y.detach().float().cpu() - ref).abs().max().item() <= 2e-4 * rms, ((y.detach().cpu() - ref).abs().max().item(), rms)
    j = torch.arange(y.numel(), dtype=torch.float64)
    ((y * torch.sin(0.37 * j).float().view_as(y).cuda()).sum() / y[0, 0].numel()).backward()
    # Gradient VALUES of this fixture are not reproducible between two fp32 implementations: the reference's own class run
    # twice with a 1e-6 perturbation of the input moves dx by more than its maximum and the parameter-gradient norms by 18 %
    # (median) - tools/probes/lightmunet_reference_conditioning.py; the forward moves 1.5e-4.  Checked here: which parameters
    # receive a gradient (the reference's list, in order), finiteness, and the overall scale of dx (its norm: 0.0682 vs 0.0684)
    rdx = torch.tensor(z["dx"])
    assert abs(x.grad.norm().item() - rdx.norm().item()) <= 0.1 * rdx.norm().item()
    names = [str(n) for n in z["names"]]
    assert [n for n, p in net.named_parameters() if p.grad is not None] == names
    assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters() if p.grad is not None)


@pytest.mark.gpu
@pytest.mark.parametrize("patch", [(128, 128), (32, 32, 32)])
def test_trainer_steps(hip_lib, patch):
    """nnUNetTrainerLightMUNet: fp32 step, one output (no deep supervision), Adam + PolyLR(0.9)"""
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training import zoo_trainers as Z
    plans, cfg, dj = nnunet_plans(len(patch), patch, batch_size=2)
    torch.manual_seed(0)
    tr = Z.nnUNetTrainerLightMUNet(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    assert type(tr.network).__name__ == "LightMUNet" and tr._get_deep_supervision_scales() is None
    assert type(tr.optimizer).__name__ == "Adam" and tr.grad_scaler is None
    b = synthetic_batch(2, patch, [[1.0] * len(patch)], seed=1)
    b = {"data": b["data"].cuda(), "target": b["target"][0].cuda()}
    before = [p.detach().clone() for p in tr.network.parameters()]
    losses = [float(tr.train_step(b)["loss"]) for _ in range(4)]
    assert all(np.isfinite(losses)), losses
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, tr.network.parameters()))
