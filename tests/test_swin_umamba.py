"""Swin-UMamba / Swin-UMamba-D (reference nets/SwinUMamba.py, nets/SwinUMambaD.py, trainers nnUNetTrainerSwinUMamba[D]) - round 4.
Pinned against the REFERENCE's own modules run on CPU (tools/make_golden_swin_umamba.py):
  * state_dict names / shapes / order and the seeded-construction digest of the whole Swin-UMamba-D factory network and of
    Swin-UMamba's VSSM encoder (tests/golden/swin_umamba_manifest.json);
  * the whole Swin-UMamba-D network - VSSM encoder, patch-expanding Mamba decoder, heads - forward, dx and parameter-gradient
    norms on a reduced configuration with the reference's parameters (tests/golden/swin_umamba_d.npz).
Unpinned (monai absent, labelled in nnuzoo_amd/nets/monai_blocks.py): the UNETR blocks of Swin-UMamba's decoder - structure,
shapes and a trainer step are tested."""
import hashlib
import json
import os
import zlib

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _pattern(shape, freq, phase):
    i = torch.arange(int(np.prod(shape)), dtype=torch.float64)
    return torch.cos(freq * i + phase).float().reshape(shape)


def _digest(sd):
    h = hashlib.sha256()
    crc = {}
    for i, (k, v) in enumerate(sd.items()):
        b = v.detach().cpu().contiguous().numpy().tobytes()
        h.update(b)
        if i % 40 == 0:
            crc[k] = zlib.crc32(b)
    return {"sha256": h.hexdigest(), "n_tensors": len(sd), "crc32": crc}


def _build(name):
    from nnuzoo_amd.nets import swin_umamba as S
    if name == "SwinUMambaD":
        return S.get_swin_umamba_d_from_plans(None, {"labels": {"a": 0, "b": 1, "c": 2}}, None, 1, deep_supervision=True,
                                              use_pretrain=False)
    return S.VSSMEncoder(patch_size=2, in_chans=48)


@pytest.mark.parametrize("name", ["SwinUMambaD", "SwinUMamba.vssm_encoder"])
def test_state_dict_and_seeded_construction_match_the_reference(name):
    want = json.load(open(os.path.join(GOLD, "swin_umamba_manifest.json")))[name]
    torch.manual_seed(0)
    net = _build(name)
    sd = net.state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == want["state_dict"]
    got = _digest(sd)
    bad = [k for k in want["seeded"]["crc32"] if got["crc32"].get(k) != want["seeded"]["crc32"][k]]
    assert not bad, bad[:5]
    assert got["sha256"] == want["seeded"]["sha256"]


def test_swin_umamba_structure_and_namespace():
    from nnuzoo_amd.nets.swin_umamba import SwinUMamba, SwinUMambaD
    from nnunetv2.nets.SwinUMamba import SwinUMamba as viaNamespace, get_swin_umamba_from_plans
    from nnunetv2.nets.SwinUMambaD import SwinUMambaD as viaNamespaceD
    from nnunetv2.training.nnUNetTrainer.nnUNetTrainerSwinUMamba import nnUNetTrainerSwinUMamba
    from nnunetv2.training.nnUNetTrainer.nnUNetTrainerSwinUMambaD import nnUNetTrainerSwinUMambaD
    from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer
    assert viaNamespace is SwinUMamba and viaNamespaceD is SwinUMambaD
    assert issubclass(nnUNetTrainerSwinUMamba, nnUNetTrainer) and issubclass(nnUNetTrainerSwinUMambaD, nnUNetTrainer)
    net = get_swin_umamba_from_plans(4, 1, deep_supervision=True, use_pretrain=False)
    keys = list(net.state_dict().keys())
    want = json.load(open(os.path.join(GOLD, "swin_umamba_manifest.json")))["SwinUMamba.vssm_encoder"]["state_dict"]
    assert keys[:4] == ["stem.0.weight", "stem.0.bias", "stem.1.weight", "stem.1.bias"]
    assert keys[4:4 + len(want)] == ["vssm_encoder." + k for k, _ in want]
    assert keys[4 + len(want)] == "encoder1.layer.conv1.conv.weight" and keys[-1] == "out_layers.3.conv.conv.bias"
    assert net.state_dict()["decoder6.transp_conv.conv.weight"].shape == (768, 768, 2, 2)
    # freeze_encoder: everything in the encoder except the patch embedding
    net.freeze_encoder()
    frozen = [n for n, p in net.named_parameters() if not p.requires_grad]
    assert frozen and all(n.startswith("vssm_encoder.") and "patch_embed" not in n for n in frozen)
    assert len(frozen) == len(want) - 4
    net.unfreeze_encoder()
    assert all(p.requires_grad for p in net.parameters())


def test_load_pretrained_ckpt_renames_downsample_layers(tmp_path):
    """VMamba checkpoints name the patch merging `layers.i.downsample`; classifier / final norm are dropped, the patch embedding
    is skipped (Swin-UMamba) or taken when its channel count matches (-D)"""
    from nnuzoo_amd.nets.swin_umamba import SwinUMambaD, load_pretrained_ckpt
    dims = [8, 16, 32, 64]
    net = SwinUMambaD(dict(in_chans=3, patch_size=4, depths=[1, 1, 1, 1], dims=dims),
                      dict(num_classes=2, deep_supervision=False, features_per_stage=dims))
    enc = {k: torch.full_like(v, 0.5) for k, v in net.vssm_encoder.state_dict().items()}
    ck = {}
    for k, v in enc.items():
        m = k.split(".")
        ck[f"layers.{m[1]}.downsample." + ".".join(m[2:]) if m[0] == "downsamples" else k] = v
    ck.update({"norm.weight": torch.zeros(64), "head.weight": torch.zeros(10, 64), "unknown.weight": torch.zeros(1)})
    path = str(tmp_path / "vmamba_tiny.pth")
    torch.save({"model": ck}, path)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    load_pretrained_ckpt(net, path, num_input_channels=3)
    after = net.state_dict()
    assert all(bool((after["vssm_encoder." + k] == 0.5).all()) for k in enc)
    assert all(torch.equal(after[k], before[k]) for k in after if k.startswith("decoder."))
    net2 = SwinUMambaD(dict(in_chans=1, patch_size=4, depths=[1, 1, 1, 1], dims=dims),
                       dict(num_classes=2, deep_supervision=False, features_per_stage=dims))
    keep = net2.state_dict()["vssm_encoder.patch_embed.proj.weight"].clone()
    load_pretrained_ckpt(net2, path, num_input_channels=1)          # 3-channel patch embedding in the file: passed over
    assert torch.equal(net2.state_dict()["vssm_encoder.patch_embed.proj.weight"], keep)
    assert bool((net2.state_dict()["vssm_encoder.layers.0.blocks.0.self_attention.out_proj.weight"] == 0.5).all())


def _reduced_d():
    from nnuzoo_amd.nets.swin_umamba import SwinUMambaD
    g = np.load(os.path.join(GOLD, "swin_umamba_d.npz"))
    dims = [int(d) for d in g["dims"]]
    net = SwinUMambaD(dict(in_chans=int(g["x"].shape[1]), patch_size=4, depths=[1, 1, 2, 1], dims=dims, drop_path_rate=0.2),
                      dict(num_classes=3, deep_supervision=True, features_per_stage=dims, drop_path_rate=0.2, d_state=16))
    return net, g


def test_reduced_network_state_dict_matches_the_reference():
    net, g = _reduced_d()
    want = [(k[2:], tuple(g[k].shape)) for k in g.files if k.startswith("p_")]
    assert [(n, tuple(p.shape)) for n, p in net.named_parameters()] == want


@pytest.mark.gpu
def test_swin_umamba_d_forward_backward_match_the_reference(hip_lib):
    net, g = _reduced_d()
    with torch.no_grad():
        for n, p in net.named_parameters():
            p.copy_(torch.from_numpy(g["p_" + n]))
    net = net.cuda().eval()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    outs = net(x)
    sens = g["sens"]
    assert len(outs) == 4
    loss = 0
    for i, o in enumerate(outs):
        ref = torch.from_numpy(g[f"out{i}"])
        assert tuple(o.shape) == tuple(ref.shape)
        scale = ref.abs().max().item()
        err = (o.detach().float().cpu() - ref).abs().max().item()
        assert err <= max(2e-4, 50 * float(sens[i])) * scale, (i, err, scale, float(sens[i]))   # fp32 end to end
        loss = loss + (o * _pattern(o.shape, 0.37, 0.5 + i).cuda()).sum() / o[0, 0].numel()
    loss.backward()
    dx, dref = x.grad.float().cpu(), torch.from_numpy(g["dx"])
    assert (dx - dref).abs().max().item() <= 1e-3 * dref.abs().max().item()
    norms = dict(zip([str(n) for n in g["grad_names"]], g["grad_norms"]))
    got = {n: float(p.grad.double().pow(2).sum().sqrt()) for n, p in net.named_parameters() if p.grad is not None}
    assert set(got) == set(norms)
    worst = max(abs(got[n] - norms[n]) / (norms[n] + 1e-6 * max(norms.values())) for n in norms)
    assert worst <= 5e-3, worst


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["SwinUMambaD", "SwinUMamba"])
def test_swin_umamba_trainer_steps_with_frozen_and_trainable_encoder(hip_lib, which):
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training import zoo_trainers
    plans, cfg, dj = nnunet_plans(2, (64, 64), batch_size=2)
    tr = getattr(zoo_trainers, "nnUNetTrainer" + which)(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    scales = tr._get_deep_supervision_scales()
    assert len(scales) == 4 and scales[1] == ([0.25, 0.25] if which == "SwinUMambaD" else [0.5, 0.5])
    b = synthetic_batch(2, (64, 64), scales, seed=3)
    enc = tr.network.vssm_encoder
    probe = enc.layers[1].blocks[0].self_attention.out_proj.weight
    embed = enc.patch_embed.proj.weight
    tr.current_epoch = 0
    tr.on_train_epoch_start()                       # epochs 0-9: encoder frozen (except the patch embedding)
    p0, e0 = probe.detach().clone(), embed.detach().clone()
    losses = [float(tr.train_step(b)["loss"]) for _ in range(4)]
    assert all(np.isfinite(l) for l in losses)
    assert torch.equal(probe.detach(), p0) and not torch.equal(embed.detach(), e0)
    tr.current_epoch = 10
    tr.on_train_epoch_start()                       # the captured step is rebuilt for the new trainable set
    losses += [float(tr.train_step(b)["loss"]) for _ in range(4)]
    assert all(np.isfinite(l) for l in losses)
    assert not torch.equal(probe.detach(), p0)
