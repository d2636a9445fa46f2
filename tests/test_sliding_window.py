"""Sliding-window inference.

CPU: tile geometry / importance map of the product's host code and the oracle's accumulation loop against fixtures
produced by the reference's own functions (tests/golden/sliding_window.npz, tools/make_golden.py gen_sliding_window).
GPU: the HIP accumulation kernels and the predictor, bit-exact against the same fixtures and against the oracle."""
import os
import types

import numpy as np
import pytest
import torch

from golden_util import toy_image, toy_seg_network
from nnuzoo_amd.inference.sliding_window_prediction import (compute_gaussian, compute_steps_for_sliding_window,
                                                              pad_to_tile)
from oracle import sliding_window as osw

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "sliding_window.npz"))
RUNS = ["3d_mirror", "3d_plain", "2d_on_3d", "3d_mirror2"]


def _run_cfg(name):
    cfg = GOLD[f"run_{name}_cfg"]
    nd_patch = 2 if name == "2d_on_3d" else 3
    dims = tuple(int(v) for v in cfg[:3])
    patch = tuple(int(v) for v in cfg[3:3 + nd_patch])
    step, gauss, seed = float(cfg[3 + nd_patch]), bool(cfg[4 + nd_patch]), int(cfg[5 + nd_patch])
    mirror = tuple(int(v) for v in GOLD[f"run_{name}_mirror"]) or None
    return dims, patch, step, gauss, seed, mirror


def test_steps_match_reference():
    for i in range(int(GOLD["n_step_cases"])):
        args = GOLD[f"steps{i}_args"]
        nd = (len(args) - 1) // 2
        img, tile, st = [int(v) for v in args[:nd]], [int(v) for v in args[nd:2 * nd]], float(args[-1])
        steps = compute_steps_for_sliding_window(img, tile, st)
        assert len(steps) == nd
        for a in range(nd):
            assert steps[a] == GOLD[f"steps{i}_axis{a}"].tolist(), (img, tile, st, a)
    with pytest.raises(AssertionError):
        compute_steps_for_sliding_window((10, 10), (8, 8), 0.0)


def test_gaussian_matches_reference():
    cpu = torch.device("cpu")
    for i in range(2):
        tile = tuple(int(v) for v in GOLD[f"gauss{i}_tile"])
        g = compute_gaussian(tile, sigma_scale=1. / 8, value_scaling_factor=10, device=cpu)
        assert g.dtype == torch.float16 and np.array_equal(g.numpy(), GOLD[f"gauss{i}"])
    g = compute_gaussian((128, 128, 128), sigma_scale=1. / 8, value_scaling_factor=10, device=cpu)
    assert np.array_equal(g[::9, ::7, ::5].numpy(), GOLD["gauss2_sample"])
    s, mn, mx = GOLD["gauss2_sum_min_max"]
    assert float(g.double().sum()) == s and float(g.min()) == mn and float(g.max()) == mx and mn > 0


def test_pad_to_tile():
    x = torch.arange(2 * 5 * 7, dtype=torch.float32).reshape(2, 5, 7)
    p, sl = pad_to_tile(x, (8, 6))
    assert p.shape == (2, 8, 7) and torch.equal(p[sl], x)
    assert p[:, 0].abs().sum() == 0 and p[:, 6:].abs().sum() == 0          # 3 extra rows: 1 below, 2 above
    p, sl = pad_to_tile(x, (4, 4))
    assert p is x and torch.equal(p[sl], x)
    x3 = torch.ones(1, 3, 4, 5)
    p, sl = pad_to_tile(x3, (4, 4, 8))
    assert p.shape == (1, 4, 4, 8) and torch.equal(p[sl], x3) and p.sum() == x3.sum()


def _slicers(dims, patch, step):
    from nnuzoo_amd.inference.predict_from_raw_data import nnUNetPredictor
    pr = nnUNetPredictor.__new__(nnUNetPredictor)
    pr.configuration_manager = types.SimpleNamespace(patch_size=list(patch))
    pr.tile_step_size = step
    return pr._internal_get_sliding_window_slicers(dims)


@pytest.mark.parametrize("name", RUNS)
def test_oracle_matches_reference_loop(name):
    dims, patch, step, gauss, seed, mirror = _run_cfg(name)
    data = toy_image(dims, seed=seed)
    slicers = _slicers(dims, patch, step)
    assert len(slicers) == int(GOLD[f"run_{name}_nslicers"])
    g = compute_gaussian(tuple(patch), sigma_scale=1. / 8, value_scaling_factor=10, device=torch.device("cpu")) \
        if gauss else None
    out = osw.predict_sliding_window(toy_seg_network, data, slicers, 2, g, mirror)
    assert np.array_equal(out.numpy(), GOLD[f"run_{name}_logits"])


def test_predictor_refuses_cpu():
    from nnuzoo_amd.inference.predict_from_raw_data import nnUNetPredictor
    with pytest.raises(RuntimeError, match="no CPU path"):
        nnUNetPredictor(device=torch.device("cpu"))


# ---- GPU ---------------------------------------------------------------------------------------------------------------
def _predictor(patch, step, gauss, mirror, tiles_per_forward=4):
    from nnuzoo_amd.inference.predict_from_raw_data import nnUNetPredictor

    class Net(torch.nn.Module):
        def forward(self, x):
            return toy_seg_network(x)

    pr = nnUNetPredictor(tile_step_size=step, use_gaussian=gauss, use_mirroring=mirror is not None,
                         device=torch.device("cuda"), allow_tqdm=False, tiles_per_forward=tiles_per_forward)
    pr.manual_initialization(Net(), None, types.SimpleNamespace(patch_size=list(patch)), None, {}, "toy", mirror,
                             label_manager=types.SimpleNamespace(num_segmentation_heads=2))
    return pr


@pytest.mark.gpu
@pytest.mark.parametrize("name", RUNS)
def test_predictor_bit_exact_vs_reference(hip_lib, name):
    dims, patch, step, gauss, seed, mirror = _run_cfg(name)
    data = toy_image(dims, seed=seed)
    for tpf in (1, 3):
        pr = _predictor(patch, step, gauss, mirror, tiles_per_forward=tpf)
        out = pr.predict_sliding_window_return_logits(data)
        assert out.dtype == torch.float16 and out.is_cuda
        assert np.array_equal(out.cpu().numpy(), GOLD[f"run_{name}_logits"])


@pytest.mark.gpu
def test_predictor_pads_small_images_and_matches_oracle(hip_lib):
    """image smaller than the tile on one axis (pad + revert), 3 mirror axes, against the CPU oracle"""
    patch, step, mirror = (16, 16, 16), 0.5, (0, 1, 2)
    data = toy_image((12, 30, 16), seed=5)
    pr = _predictor(patch, step, True, mirror)
    out = pr.predict_sliding_window_return_logits(data)
    padded, sl = pad_to_tile(data, patch)
    g = compute_gaussian(patch, sigma_scale=1. / 8, value_scaling_factor=10, device=torch.device("cpu"))
    ref = osw.predict_sliding_window(toy_seg_network, padded, _slicers(padded.shape[1:], patch, step), 2, g, mirror)
    ref = ref[(slice(None), *sl[1:])]
    assert out.shape == ref.shape == (2, 12, 30, 16)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.gpu
def test_accumulate_kernel_random_halves(hip_lib):
    """the kernel alone on arbitrary half data (rounding paths the toy network does not reach): two overlapping tiles
    with all 8 mirror variants, bit-exact against torch's half arithmetic"""
    import itertools
    g = torch.Generator().manual_seed(3)
    pr = _predictor((8, 12, 10), 0.5, True, (0, 1, 2))
    combos = pr._mirror_axes_combinations(3)
    image = (12, 12, 16)
    gauss = (torch.rand(8, 12, 10, generator=g) * 10 + 0.01).half()
    logits = torch.zeros((3, *image), dtype=torch.half)
    npred = torch.zeros(image, dtype=torch.half)
    d_logits, d_npred = logits.cuda(), npred.cuda()
    for origin in [(0, 0, 0), (4, 0, 6), (2, 0, 3)]:
        preds = (torch.randn(8, 3, 8, 12, 10, generator=g) * 3).half()
        pr._accumulate(preds.cuda(), combos, gauss.cuda(), d_logits, d_npred, origin)
        p = preds[0].clone()
        for m, c in enumerate(combos[1:], start=1):
            p += torch.flip(preds[m], [a + 1 for a in c])
        p /= len(combos)
        p *= gauss
        sl = tuple(slice(o, o + t) for o, t in zip(origin, (8, 12, 10)))
        logits[(slice(None), *sl)] += p
        npred[sl] += gauss
    torch.cuda.synchronize()
    assert torch.equal(d_logits.cpu(), logits) and torch.equal(d_npred.cpu(), npred)


@pytest.mark.gpu
def test_predictor_with_hip_network_matches_oracle_loop(hip_lib):
    """the real hot path as the tile network: nnuzoo_amd.PlainConvUNet (HIP schedule) under the predictor (mirror
    variants batched into one forward) against the oracle loop calling the same network tile by tile.  The network's
    InstanceNorm statistics are summed with fp32 atomics, so logits agree to fp16 rounding, not bit for bit."""
    from oracle.plain_conv_unet import planner_arch_kwargs
    from nnuzoo_amd.inference.predict_from_raw_data import nnUNetPredictor
    from nnuzoo_amd.nets.plain_conv_unet import PlainConvUNet
    from nnuzoo_amd.utilities.network_initialization import InitWeights_He
    torch.manual_seed(0)
    net = PlainConvUNet(1, num_classes=3, **planner_arch_kwargs(3, 3, [32, 64, 128], deep_supervision=False))
    net.apply(InitWeights_He(1e-2))
    net = net.cuda().eval()
    patch, mirror = (16, 32, 32), (0, 1, 2)
    pr = nnUNetPredictor(tile_step_size=0.5, use_gaussian=True, use_mirroring=True, device=torch.device("cuda"),
                         allow_tqdm=False)
    pr.manual_initialization(net, None, types.SimpleNamespace(patch_size=list(patch)), None, {}, "nnUNetTrainer", mirror,
                             label_manager=types.SimpleNamespace(num_segmentation_heads=3))
    data = torch.randn(1, 20, 40, 36, generator=torch.Generator().manual_seed(1))
    out = pr.predict_sliding_window_return_logits(data)
    assert out.shape == (3, 20, 40, 36)

    def tile_net(x):
        with torch.no_grad():
            return net(x.cuda()).cpu()

    g = compute_gaussian(patch, sigma_scale=1. / 8, value_scaling_factor=10, device=torch.device("cpu"))
    ref = osw.predict_sliding_window(tile_net, data, _slicers(data.shape[1:], patch, 0.5), 3, g, mirror)
    o, r = out.float().cpu(), ref.float()
    # Where the summed importance weight is in fp16's subnormal range (image corners: only the far tail of one tile's
    # gaussian) the reference's half accumulators hold 1-3 significant bits, so two network outputs that differ by
    # one fp16 ulp give visibly different quotients - in the reference as here.  Compare where the weight is normal.
    wsum = torch.zeros(data.shape[1:])
    for sl in _slicers(data.shape[1:], patch, 0.5):
        wsum[sl[1:]] += g.float()
    ok = (wsum > 1e-3)[None].expand_as(o)
    assert ok.float().mean().item() > 0.7, ok.float().mean().item()
    err = ((o - r).abs() * ok).max().item()
    assert err <= 1e-2 * r.abs().max().item() + 1e-2 * (r.abs() * ok).max().item(), err
    agree = ((o.argmax(0) == r.argmax(0)) | ~ok[0]).float().mean().item()
    assert agree > 0.999
