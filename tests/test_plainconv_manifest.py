"""PlainConvUNet against what the reference's in-tree code pins (tests/golden/plainconv_manifest.json, written by
tools/make_plainconv_manifest.py from the reference's own `get_pool_and_conv_props` and planner constants): topology,
constructor kwargs, parameter counts (31.20 M for 3d_fullres 128^3, 46.32 M for 2d 512^2), deep-supervision output
shapes / scales and the per-layer forward GFLOP list of SURVEY.md 8d.  Both the CPU oracle and the product's
parameter-holder tree + execution plan are asserted against it.  The wiring inside the third-party class stays
"parity unpinned"  - this file pins everything else.  CPU only."""
import json
import os
import pydoc

import numpy as np
import pytest
import torch

from nnuzoo_amd.nets.plain_conv_unet import PlainConvUNet, _Plan, _to3
from nnuzoo_amd.synthetic import conv_flops_forward, nnunet_plans
from oracle.plain_conv_unet import OraclePlainConvUNet

HERE = os.path.dirname(os.path.abspath(__file__))
DOC = json.load(open(os.path.join(HERE, "golden", "plainconv_manifest.json")))
CASES = {c["name"]: c for c in DOC["cases"]}


def _kwargs(case):
    kw = dict(case["arch_kwargs"])
    for k in ("conv_op", "norm_op", "dropout_op", "nonlin"):     # get_network_from_plans.py:31-35
        if isinstance(kw[k], str):
            kw[k] = pydoc.locate(kw[k])
    return kw


def test_manifest_headline_numbers():
    """SURVEY.md 8a / 8d figures, now derived from the reference's own topology function"""
    c3, c2 = CASES["3d_fullres_128"], CASES["2d_512"]
    assert c3["arch_kwargs"]["features_per_stage"] == [32, 64, 128, 256, 320, 320]
    assert c3["arch_kwargs"]["strides"] == [[1, 1, 1]] + [[2, 2, 2]] * 5
    assert round(c3["parameter_count"] / 1e6, 2) == 31.20 and round(c2["parameter_count"] / 1e6, 2) == 46.32
    assert abs(c3["forward_gflop_per_sample_total"] - 954.6) < 0.1
    assert abs(c2["forward_gflop_per_sample_total"] - 119.2) < 0.1
    assert abs(c3["forward_gflop_per_sample"]["dec0.0"] - 231.93) < 0.01
    assert c3["deep_supervision_output_shapes"] == [[2, 128 >> i, 128 >> i, 128 >> i] for i in range(5)]
    assert len(c2["deep_supervision_output_shapes"]) == 7 and c2["deep_supervision_output_shapes"][-1] == [2, 8, 8]
    assert DOC["sources"]["planner_constants"]["UNet_featuremap_min_edge_length"] == 4


@pytest.mark.parametrize("name", ["3d_fullres_128", "2d_512", "3d_small_64"])
def test_synthetic_planner_reproduces_the_reference_topology(name):
    """bench.py / the GPU tests build their plans with nnuzoo_amd.synthetic.nnunet_plans: same kwargs as the reference"""
    c = CASES[name]
    plans, cfg, _ = nnunet_plans(len(c["patch_size"]), c["patch_size"], num_classes=c["num_classes"])
    arch = plans["configurations"][cfg]["architecture"]
    assert arch["network_class_name"] == "dynamic_network_architectures.architectures.unet.PlainConvUNet"
    got = dict(arch["arch_kwargs"])
    want = dict(c["arch_kwargs"])
    for k in want:
        assert got[k] == want[k] or list(got[k]) == list(want[k]), (k, got[k], want[k])
    fl = conv_flops_forward(got, c["patch_size"], c["input_channels"], c["num_classes"])
    assert fl.keys() == c["forward_gflop_per_sample"].keys()
    for k, v in c["forward_gflop_per_sample"].items():
        assert abs(fl[k] / 1e9 - v) <= 1e-6 * max(1.0, v), k


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_and_product_match_the_manifest(name, hip_lib):
    c = CASES[name]
    kw = _kwargs(c)
    nd = len(c["patch_size"])
    oracle = OraclePlainConvUNet(c["input_channels"], num_classes=c["num_classes"], deep_supervision=True, **kw)
    net = PlainConvUNet(c["input_channels"], num_classes=c["num_classes"], deep_supervision=True, **kw)
    n_oracle = sum(p.numel() for p in oracle.parameters())
    n_net = sum(p.numel() for p in net.parameters())
    assert n_oracle == n_net == c["parameter_count"]
    assert [(k, tuple(v.shape)) for k, v in oracle.state_dict().items()] == \
        [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    # the product's execution plan (host-side: no launch): level geometry and output shapes at the manifest's patch size
    plan = _Plan(net, 2, _to3(c["patch_size"]))
    shapes = [[c["num_classes"]] + list(plan.level_dims[lvl][3 - nd:]) for lvl in range(plan.S - 1)]
    assert shapes == c["deep_supervision_output_shapes"]
    # the trainer's deep-supervision scales come from the strides (nnUNetTrainer.py:401-408)
    scales = list(list(i) for i in 1 / np.cumprod(np.vstack(c["arch_kwargs"]["strides"]), axis=0))[:-1]
    assert np.allclose(np.array(scales), np.array(c["deep_supervision_scales"]))
    # algorithmic FLOPs bench.py prices the conv kernels with = the manifest's per-layer figures (batch 2)
    mf = c["forward_gflop_per_sample"]
    enc = [b for blocks in plan.enc_blocks for b in blocks]
    names = [f"enc{s}.{i}" for s, blocks in enumerate(plan.enc_blocks) for i in range(len(blocks))]
    for b, nme in zip(enc, names):
        if b.stem:
            continue
        algo = 2.0 * b.V * b.cin_w * b.cout * b.nk / 1e9
        assert abs(algo - mf[nme]) <= 1e-6 * mf[nme], nme
        if not b.padded:
            assert abs(b.fwd.flops / 2 / 1e9 - mf[nme]) <= 1e-6 * mf[nme], nme
    for j, blocks in enumerate(plan.dec_blocks):
        lvl = plan.S - 2 - j
        assert abs(plan.ups[j].fwd.flops / 2 / 1e9 - mf[f"up{lvl}"]) <= 1e-6 * mf[f"up{lvl}"]
        for i, b in enumerate(blocks):
            assert abs(b.fwd.flops / 2 / 1e9 - mf[f"dec{lvl}.{i}"]) <= 1e-6 * mf[f"dec{lvl}.{i}"]


def test_oracle_forward_output_order_small():
    """highest resolution first, one logit map per decoder stage (nnUNetTrainer.py:1010-1022): oracle run on CPU at a
    size it finishes in a second, shapes against the manifest's rule (level l = patch / prod(strides[:l+1]))"""
    c = CASES["3d_small_64"]
    kw = _kwargs(c)
    oracle = OraclePlainConvUNet(1, num_classes=2, deep_supervision=True, **kw).eval()
    with torch.no_grad():
        outs = oracle(torch.zeros(1, 1, 64, 64, 64))
    assert [list(o.shape[1:]) for o in outs] == c["deep_supervision_output_shapes"]
    oracle.decoder.deep_supervision = False
    with torch.no_grad():
        o = oracle(torch.zeros(1, 1, 64, 64, 64))
    assert list(o.shape[1:]) == c["deep_supervision_output_shapes"][0]
