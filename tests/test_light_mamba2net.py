"""LightMamba2Net / Mamba2 (nnuzoo_amd/nets/light_mamba2net.py, mamba2.py; reference nets/light_mamba2net.py).

Pinned part: the Mamba2 mixer - against tests/golden/mamba2_mixer.npz, outputs and gradients of HuggingFace transformers'
independent pure-torch Mamba2Mixer (tools/make_mamba2_golden.py; mamba_ssm itself is absent from the reference tree).
CPU: the oracle restatement (oracle/mamba2.py) against that golden; GPU: the product module (HIP scan / conv1d / gate
kernels) against the golden.  The net around it (monai helper layers) is PARITY UNPINNED: structure, scale tables taken
from the reference's formulas, shapes and finiteness, and a trainer step."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden", "mamba2_mixer.npz")
NAMES = ["in_proj.weight", "conv1d.weight", "conv1d.bias", "dt_bias", "A_log", "D", "norm.weight", "out_proj.weight"]


def _close(a, b, tol):
    b = torch.as_tensor(b, dtype=torch.float32)
    return torch.allclose(a.detach().float().cpu(), b, rtol=tol, atol=tol * max(1e-6, b.abs().max().item()))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_mixer_matches_hf_golden(tag):
    from oracle.mamba2 import mamba2_mixer
    z = np.load(G)
    p = {n: torch.tensor(z[f"{tag}.p.{n}"]).double().requires_grad_() for n in NAMES}
    u = torch.tensor(z[f"{tag}.u"]).double().requires_grad_()
    out = mamba2_mixer(u, p, int(z[f"{tag}.headdim"]))
    (out * torch.tensor(z[f"{tag}.gout"]).double()).sum().backward()
    assert _close(out, z[f"{tag}.out"], 2e-6) and _close(u.grad, z[f"{tag}.du"], 2e-6)
    for n in NAMES:
        assert _close(p[n].grad, z[f"{tag}.g.{n}"], 2e-6), n


def test_mixer_parameter_names_shapes_and_headdim_rule():
    from nnuzoo_amd.nets.light_mamba2net import MambaLayer
    from nnuzoo_amd.nets.mamba2 import Mamba2
    z = np.load(G)
    for tag, d_model in (("a", 16), ("b", 64)):
        hd = MambaLayer.get_nheaddim(d_model, 2)
        assert hd == int(z[f"{tag}.headdim"])
        m = Mamba2(d_model, d_state=16, d_conv=4, expand=2, headdim=hd)
        assert [n for n, _ in m.named_parameters()] == ["dt_bias", "A_log", "D", "in_proj.weight", "conv1d.weight",
                                                        "conv1d.bias", "norm.weight", "out_proj.weight"]
        for n, p in m.named_parameters():
            assert tuple(p.shape) == z[f"{tag}.p.{n}"].shape, n
        assert bool((m.dt_bias.detach() < 0).all()) and bool((m.A_log.detach() >= 0).all()) and bool((m.D == 1).all())
    assert [MambaLayer.get_nheaddim(d, 2) for d in (16, 32, 64, 128, 256)] == [2, 4, 8, 16, 32]   # 16 heads at every width
    with pytest.raises(RuntimeError):
        Mamba2(16, d_state=16, headdim=2)(torch.zeros(1, 8, 16))                                   # no CPU path


def test_get_scales_min_size_rule():
    from nnuzoo_amd.nets.ssnd2net import get_scales
    # light_mamba2net.py:562-600: halve unless odd or the half would drop below min_size
    assert get_scales(2, (512, 512), 5, None, min_size=8) == [(2, 2)] * 5
    assert get_scales(2, (64, 64), 5, None, min_size=8) == [(2, 2)] * 3 + [(1, 1)] * 2
    assert get_scales(3, (32, 64, 64), 5, None, min_size=8) == [(2, 2, 2), (2, 2, 2), (1, 2, 2), (1, 1, 1), (1, 1, 1)]
    assert get_scales(2, (16, 16), 3, min_size=4) == [(2, 2), (2, 2), (1, 1)]
    assert get_scales(2, (320, 192), 5, None) == [(2, 2)] * 5                                        # min_size 1: as before
    assert get_scales(2, (40, 24), 4, None) == [(2, 2), (2, 2), (2, 2), (1, 1)]                       # odd sizes stop


def test_whole_net_structure_cpu():
    from nnuzoo_amd.nets.light_mamba2net import LightMamba2Net, LightMamba2NetP, LightMUNet
    net = LightMamba2Net(2, 1, 2, True, [512, 512])
    keys = set(net.state_dict())
    for k in ("stage1.convInit.0.conv.weight", "stage1.down_layers.1.1.gsc.proj.1.conv.bias",
              "stage1.down_layers.0.1.mamba1.mamba.dt_bias", "stage1.down_layers.6.2.mamba2.mamba.norm.weight",
              "stage3.down_layers.2.1.mamba1.skip_scale", "stage2d.up_layers.0.0.skip_scale",
              "stage4.up_samples.0.0.conv.weight", "stage1.conv_final.2.1.conv.bias", "patch_merging5.reduction.weight",
              "patch_expand5d.expand.weight", "concat_back_dim4d.weight", "side6.conv.weight", "outconv.conv.bias"):
        assert k in keys, k
    s1 = net.stage1
    assert isinstance(s1, LightMUNet) and s1.blocks_down == [1] + [2] * 6 and s1.blocks_up == [1] * 6
    assert s1.scales == [(1, 1)] + [(2, 2)] * 6 and net.stage6.scales == [(1, 1), (2, 2), (2, 2), (1, 1)]
    assert [blk[1].order for blk in s1.down_layers] == ['h w', 'w h'] * 3 + ['h w']
    assert s1.down_layers[0][1].mamba1.mamba.nheads == 16 and s1.down_layers[0][1].mamba1.mamba.headdim == 2
    small = LightMamba2NetP(2, 1, 2, True, [512, 512])
    assert isinstance(small.concat_back_dim4d, torch.nn.Identity) and small.side1.conv.kernel_size == (3, 3)
    assert sum(p.numel() for p in small.parameters()) < 0.1 * sum(p.numel() for p in net.parameters())
    net3 = LightMamba2Net(3, 1, 3, False, [32, 64, 64])
    assert [blk[1].order for blk in net3.stage1.down_layers][:4] == ['d h w', 'd w h', 'w h d', 'd h w']


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_mixer_forward_backward_vs_hf_golden(hip_lib, tag):
    from nnuzoo_amd.nets.mamba2 import Mamba2
    z = np.load(G)
    d_model = z[f"{tag}.u"].shape[-1]
    m = Mamba2(d_model, d_state=16, d_conv=4, expand=2, headdim=int(z[f"{tag}.headdim"]))
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.from_numpy(z[f"{tag}.p.{n}"]))
    m = m.cuda()
    u = torch.from_numpy(z[f"{tag}.u"]).cuda().requires_grad_(True)
    out = m(u)
    (out * torch.from_numpy(z[f"{tag}.gout"]).cuda()).sum().backward()
    assert _close(out, z[f"{tag}.out"], 2e-4) and _close(u.grad, z[f"{tag}.du"], 5e-4)
    for n, p in m.named_parameters():
        assert _close(p.grad, z[f"{tag}.g.{n}"], 1e-3), n


@pytest.mark.gpu
@pytest.mark.parametrize("small", [False, True])
def test_whole_net_forward_backward_and_trainer_step(hip_lib, small):
    from nnuzoo_amd.nets.light_mamba2net import LightMamba2Net, LightMamba2NetP
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training import zoo_trainers as Z
    torch.manual_seed(0)
    prev = torch.backends.cudnn.enabled
    torch.backends.cudnn.enabled = False          # as the trainer: ATen's native small-channel fp32 convolutions
    try:
        net = (LightMamba2NetP if small else LightMamba2Net)(2, 1, 2, True, [64, 64]).cuda()
        outs = net(torch.randn(2, 1, 64, 64, device="cuda"))
        assert [tuple(o.shape[2:]) for o in outs] == [(64, 64), (64, 64), (32, 32), (16, 16), (8, 8), (8, 8), (8, 8)]
        sum(o.float().pow(2).mean() for o in outs).backward()
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
        del net, outs
        plans, cfg, dj = nnunet_plans(2, (64, 64), batch_size=2)
        tr = getattr(Z, "nnUNetTrainerLightMamba2Net" + ("P" if small else ""))(plans, cfg, 0, dj, device=torch.device("cuda"))
        tr.initialize()
        scales = tr._get_deep_supervision_scales()
        assert scales == [[1.0, 1.0], [1.0, 1.0], [0.5, 0.5], [0.25, 0.25], [0.125, 0.125], [0.125, 0.125], [0.125, 0.125]]
        b = synthetic_batch(2, (64, 64), scales, seed=1)
        losses = [float(tr.train_step({"data": b["data"], "target": b["target"]})["loss"]) for _ in range(3)]
        assert all(np.isfinite(losses))
    finally:
        torch.backends.cudnn.enabled = prev
