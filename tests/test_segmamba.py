"""SegMamba (reference nets/seg_mamba/segmamba.py, trainer nnUNetTrainerSegMamba) - round 4.
Pinned: the whole `MambaEncoder` (stem / down-sampling convolutions, GSC, MambaLayer with the bimamba v3 / v2 block, MlpChannel)
against outputs, dx and parameter-gradient norms of the REFERENCE's own module run on CPU (tools/make_golden_segmamba.py,
tests/golden/segmamba_encoder_{3d,2d}.npz; parameters = the reference's seeded construction, stored in the fixture).
Unpinned (monai absent, labelled in nnuzoo_amd/nets/monai_blocks.py): the UNETR-style encoder / decoder blocks around it -
structure, shapes and a trainer step are tested."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = [("3d", 3), ("2d", 2)]


def _pattern(shape, freq, phase):
    i = torch.arange(int(np.prod(shape)), dtype=torch.float64)
    return torch.cos(freq * i + phase).float().reshape(shape)


def _encoder(tag, sd):
    from nnuzoo_amd.nets.segmamba import MambaEncoder
    g = np.load(os.path.join(GOLD, f"segmamba_encoder_{tag}.npz"))
    enc = MambaEncoder(spatial_dims=sd, in_chans=int(g["x"].shape[1]), depths=[2, 2, 2, 2], dims=[int(d) for d in g["dims"]])
    return enc, g


@pytest.mark.parametrize("tag,sd", CASES)
def test_encoder_state_dict_matches_the_reference(tag, sd):
    """names, order and shapes of every parameter of the reference's MambaEncoder (the fixture stores them as p_<name>)"""
    enc, g = _encoder(tag, sd)
    want = [(k[2:], tuple(g[k].shape)) for k in g.files if k.startswith("p_")]
    mine = [(n, tuple(p.shape)) for n, p in enc.named_parameters()]
    assert mine == want


def test_segmamba_structure_and_namespace():
    from nnuzoo_amd.nets.segmamba import SegMamba
    from nnunetv2.nets.seg_mamba.segmamba import SegMamba as viaNamespace
    from nnunetv2.training.nnUNetTrainer.nnUNetTrainerSegMamba import nnUNetTrainerSegMamba
    from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer
    assert viaNamespace is SegMamba and issubclass(nnUNetTrainerSegMamba, nnUNetTrainer)
    net = SegMamba(in_ch=1, out_ch=3, spatial_dims=3)
    keys = list(net.state_dict().keys())
    assert keys[0] == "vit.downsample_layers.0.0.conv.weight" and "vit.stages.3.1.mamba.out_proj.weight" in keys
    assert "vit.stages.0.0.mamba.A_s_log" in keys and "vit.gscs.2.proj4.conv.bias" in keys and keys[-1] == "out.conv.conv.bias"
    assert sum(p.numel() for p in net.parameters()) == 67362723


@pytest.mark.gpu
@pytest.mark.parametrize("tag,sd", CASES)
def test_encoder_forward_backward_match_the_reference(hip_lib, tag, sd):
    enc, g = _encoder(tag, sd)
    with torch.no_grad():
        for n, p in enc.named_parameters():
            p.copy_(torch.from_numpy(g["p_" + n]))
    enc = enc.cuda().eval()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    outs = enc(x)
    sens = g["sens"]
    loss = 0
    for i, o in enumerate(outs):
        ref = torch.from_numpy(g[f"out{i}"])
        scale = ref.abs().max().item()
        err = (o.detach().float().cpu() - ref).abs().max().item()
        # the reference's own response to a 1e-6 input perturbation (sens) bounds what two fp32 implementations can agree to
        assert err <= max(2e-3, 50 * float(sens[i])) * scale, (i, err, scale, float(sens[i]))
        loss = loss + (o * _pattern(o.shape, 0.37, 0.5 + i).cuda()).sum() / o[0, 0].numel()
    loss.backward()
    dx, dref = x.grad.float().cpu(), torch.from_numpy(g["dx"])
    tol = max(5e-3, 100 * float(sens.max()))
    assert (dx - dref).abs().max().item() <= tol * dref.abs().max().item()
    norms = dict(zip([str(n) for n in g["grad_names"]], g["grad_norms"]))
    got = {n: float(p.grad.double().pow(2).sum().sqrt()) for n, p in enc.named_parameters() if p.grad is not None}
    assert set(got) == set(norms)
    worst = max(abs(got[n] - norms[n]) / (norms[n] + 1e-6 * max(norms.values())) for n in norms)
    assert worst <= max(2e-2, 200 * float(sens.max())), worst


@pytest.mark.gpu
def test_segmamba_trainer_step_3d(hip_lib):
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerSegMamba
    plans, cfg, dj = nnunet_plans(3, (32, 32, 32), batch_size=1)
    tr = nnUNetTrainerSegMamba(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    assert tr._get_deep_supervision_scales() is None
    b = synthetic_batch(1, (32, 32, 32), [[1.0] * 3], seed=3)
    b = {"data": b["data"], "target": b["target"][0]}
    losses = [float(tr.train_step(b)["loss"]) for _ in range(3)]
    assert all(np.isfinite(l) for l in losses)
    assert sum(p.grad is not None for p in tr.network.vit.parameters()) > 100
