"""CPU: the zoo models expose exactly the reference's state_dict (names, shapes AND order), recorded from the
reference's own classes by tools/make_golden.py -> tests/golden/state_dict_manifest.json."""
import json
import os

import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["M2Net", "M2NetP", "SwT2Net"])
def test_state_dict_manifest(name):
    from nnuzoo_amd.nets import m2net, swt2net
    cls = {"M2Net": m2net.M2Net, "M2NetP": m2net.M2NetP, "SwT2Net": swt2net.SwT2Net}[name]
    net = cls(1, 2, True)
    mine = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    ref = json.load(open(os.path.join(G, "state_dict_manifest.json")))[name]
    assert len(mine) == len(ref)
    assert mine == ref


def test_plain_conv_unet_keys_follow_dna_layout():
    """dynamic_network_architectures 0.3.x naming (unverifiable here): spot-check the expected keys."""
    from torch import nn
    from nnuzoo_amd.nets.plain_conv_unet import PlainConvUNet
    from oracle.plain_conv_unet import planner_arch_kwargs
    net = PlainConvUNet(1, num_classes=2, **planner_arch_kwargs(3, 3, [32, 64, 128]))
    keys = set(net.state_dict().keys())
    for k in ["encoder.stages.0.0.convs.0.conv.weight", "encoder.stages.0.0.convs.0.norm.bias",
              "encoder.stages.0.0.convs.0.all_modules.0.weight", "encoder.stages.2.0.convs.1.all_modules.1.weight",
              "decoder.encoder.stages.1.0.convs.0.conv.bias", "decoder.stages.0.convs.0.conv.weight",
              "decoder.transpconvs.1.weight", "decoder.seg_layers.0.bias"]:
        assert k in keys, k
    assert net.state_dict()["decoder.transpconvs.0.weight"].shape == (128, 64, 2, 2, 2)
    assert net.state_dict()["decoder.stages.0.convs.0.conv.weight"].shape == (64, 128, 3, 3, 3)
