"""CPU tests: the oracle restatements (oracle/) against the golden vectors produced from the reference's own code
(tests/golden/*.npz, tools/make_golden.py) and the known-answer values of SURVEY.md §8c."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import losses as OL
from oracle.selective_scan import selective_scan_loop, selective_scan_torch

G = os.path.join(os.path.dirname(__file__), "golden")


def test_scan_kat1():
    u = np.linspace(-1, 1, 8, dtype=np.float32).reshape(1, 2, 4)
    delta = np.linspace(.1, .8, 8, dtype=np.float32).reshape(1, 2, 4)
    A = -np.array([[1., 2.], [1., 2.]], np.float32)
    B = np.linspace(.5, 1.5, 8, dtype=np.float32).reshape(1, 1, 2, 4)
    C = np.linspace(1, -1, 8, dtype=np.float32).reshape(1, 1, 2, 4)
    y = selective_scan_loop(u, delta, A, B, C, np.array([1., .5]), np.array([0., -.5]), True)
    kat = [-1.2582601, -0.7297925, -0.1838676, 0.1108990, 0.1057828, 0.2012926, -0.0189671, -0.7899251]
    assert np.allclose(y.flatten(), kat, atol=2e-7)
    assert np.allclose(np.load(os.path.join(G, "selective_scan_kat1.npz"))["y"].flatten(), kat, atol=2e-7)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "selective_scan_[a-d].npz"))))
def test_scan_oracle_vs_reference_golden(path):
    z = np.load(path)
    y = selective_scan_loop(z["u"], z["delta"], z["A"], z["B"], z["C"], z["D"], z["delta_bias"], True)
    assert np.allclose(y, z["y"], rtol=2e-5, atol=2e-5)
    t = {k: torch.tensor(z[k], requires_grad=True) for k in ["u", "delta", "A", "B", "C", "D", "delta_bias"]}
    yt = selective_scan_torch(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["delta_bias"], True)
    assert torch.allclose(yt, torch.tensor(z["y"]), rtol=2e-5, atol=2e-5)
    grads = torch.autograd.grad(yt, list(t.values()), torch.tensor(z["dy"]))
    for name, g in zip(["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"], grads):
        ref = torch.tensor(z[name])
        assert torch.allclose(g, ref, rtol=1e-4, atol=1e-4 * ref.abs().max().item()), name


def test_loss_kats():
    logits = torch.stack([torch.linspace(-2, 2, 32).view(2, 4, 4), torch.linspace(1, -1, 32).view(2, 4, 4)], 1)
    target = (torch.arange(32).view(2, 1, 4, 4) % 3 == 0).to(torch.int16)
    assert abs(float(OL.dc_and_ce(logits, target, True)) - 0.5730621) < 1e-6
    outs = [logits, logits[..., ::2, ::2], logits[..., ::4, ::4]]
    tg = [target, target[..., ::2, ::2], target[..., ::4, ::4]]
    assert abs(float(OL.deep_supervision_loss(outs, tg, True, [2 / 3, 1 / 3, 0])) - 0.6382819) < 1e-6


@pytest.mark.parametrize("tag", ["2d", "3d"])
def test_loss_oracle_vs_reference_golden(tag):
    z = np.load(os.path.join(G, f"loss_{tag}.npz"))
    logits = torch.tensor(z["logits"], requires_grad=True)
    target = torch.tensor(z["target"])
    l = OL.dc_and_ce(logits, target, bool(z["batch_dice"]))
    assert abs(float(l) - float(z["loss"])) < 1e-6
    (g,) = torch.autograd.grad(l, logits)
    assert torch.allclose(g, torch.tensor(z["dlogits"]), atol=1e-7, rtol=1e-5)
    nd = logits.dim() - 2
    sl = (..., *([slice(None, None, 2)] * nd))
    outs = [logits.detach(), logits.detach()[sl]]
    tgs = [target, target[sl]]
    outs.append(outs[1][sl])
    tgs.append(tgs[1][sl])
    lds = OL.deep_supervision_loss(outs, tgs, bool(z["batch_dice"]), z["ds_weights"])
    assert abs(float(lds) - float(z["ds_loss"])) < 1e-6
    assert np.allclose(OL.ds_weights(3), z["ds_weights"])


@pytest.mark.parametrize("tag", ["2d", "3d"])
def test_loss_oracle_ignore_label_vs_reference_golden(tag):
    """DC_and_CE_loss(ignore_label=C): fixtures from the reference's own module (tools/make_golden.py gen_losses)"""
    O = OL
    g = np.load(os.path.join(G, f"loss_{tag}.npz"))
    gi = np.load(os.path.join(G, f"loss_ignore_{tag}.npz"))
    x = torch.from_numpy(g["logits"]).requires_grad_(True)
    t = torch.from_numpy(gi["target"])
    ig = int(gi["ignore_label"])
    l = O.dc_and_ce(x, t, bool(g["batch_dice"]), ignore_label=ig)
    (gr,) = torch.autograd.grad(l, x)
    assert abs(float(l) - float(gi["loss"])) < 1e-6
    assert np.allclose(gr.numpy(), gi["dlogits"], atol=1e-7)
    l_all = O.dc_and_ce(x.detach(), torch.full_like(t, ig), bool(g["batch_dice"]), ignore_label=ig)
    assert abs(float(l_all) - float(gi["loss_all_ignored"])) < 1e-6


@pytest.mark.parametrize("tag", ["2d", "3d"])
def test_region_loss_oracle_vs_reference_golden(tag):
    """DC_and_BCE_loss (region-based training), plain and with the ignore-mask channel (tools/make_golden.py)"""
    g = np.load(os.path.join(G, f"loss_regions_{tag}.npz"))
    x, r, ig = torch.from_numpy(g["logits"]), torch.from_numpy(g["regions"]), torch.from_numpy(g["ignore"])
    for name, use, t in (("plain", False, r), ("masked", True, torch.cat([r, ig], 1))):
        xx = x.clone().requires_grad_(True)
        l = OL.dc_and_bce(xx, t, bool(g["batch_dice"]), use)
        (gr,) = torch.autograd.grad(l, xx)
        assert abs(float(l.detach()) - float(g[f"{name}_loss"])) < 1e-6
        assert np.allclose(gr.numpy(), g[f"{name}_dlogits"], atol=1e-7)


@pytest.mark.parametrize("tag", ["2d", "3d"])
def test_tp_fp_fn_oracle_vs_reference_golden(tag):
    """oracle.losses.tp_fp_fn_hard / region_tp_fp_fn against the outputs of the reference's own get_tp_fp_fn_tn driven
    as validation_step drives it (tests/golden/tp_fp_fn.npz, tools/make_golden.py:gen_tp_fp_fn); exact counts"""
    z = np.load(os.path.join(G, "tp_fp_fn.npz"))
    logits = torch.from_numpy(z[f"{tag}_logits"])
    got = torch.stack(OL.tp_fp_fn_hard(logits, torch.from_numpy(z[f"{tag}_target"]))).numpy()
    assert np.array_equal(got, z[f"{tag}_plain"])
    assert got[0].sum() + got[2].sum() == z[f"{tag}_target"].size          # every voxel is a TP or an FN of its class
    got = torch.stack(OL.tp_fp_fn_hard(logits, torch.from_numpy(z[f"{tag}_target_ignore"]),
                                       int(z[f"{tag}_ignore_label"]))).numpy()
    assert np.array_equal(got, z[f"{tag}_ignore"])
    r, ig = torch.from_numpy(z[f"{tag}_regions"]), torch.from_numpy(z[f"{tag}_regions_ignore_channel"])
    assert np.array_equal(torch.stack(OL.region_tp_fp_fn(logits, r, False)).numpy(), z[f"{tag}_regions_plain"])
    assert np.array_equal(torch.stack(OL.region_tp_fp_fn(logits, torch.cat([r, ig], 1), True)).numpy(),
                          z[f"{tag}_regions_masked"])
