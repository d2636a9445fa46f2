"""Round-3 fixtures for LM2Net / LM2NetP (nets/lm2net.py), generated in the BUILD CONTAINER from the reference's own module
imported under tools/ref_shim.py; only arrays / shapes are stored (tests/golden/), no reference source travels.
    python tools/make_golden_lm2net.py
Bindings of the absent third-party names (on top of ref_shim.install()):
  mamba_ssm.Mamba                    -> the reference's vendored block (nets/seg_mamba/mamba_simple.py, bimamba "none", slow
                                        path on the reference's selective_scan_ref) - same mathematics as mamba_ssm's block
  monai get_conv_layer(k=1)          -> ref_shim.Convolution(conv_only, bias False)          } monai is absent: these four
  monai get_upsample_layer           -> nn.Upsample(scale_factor, bi/trilinear, align_corners False) } restate its semantics
  monai get_norm_layer(("GROUP", kw)) -> nn.GroupNorm;  get_act_layer(("RELU", kw)) -> nn.ReLU      } (PARITY UNPINNED for them)
Outputs: net_LM2NetP_64.npz / net_LM2Net_64.npz (7 outputs; eval mode, BatchNorm running statistics reset to 0 / 1),
netgrad_*.npz (dx + L2 norm and 256 strided samples of every parameter gradient), lm2net_manifest.json (state_dict names /
shapes / order)."""
import json
import os
import sys

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_shim  # noqa: E402
from golden_util import det_fill  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def bind():
    ref = ref_shim.install()
    import monai.networks.blocks.convolutions as mc
    mc.Convolution = ref_shim.Convolution
    import monai.networks.blocks.segresnet_block as sb
    sb.get_conv_layer = lambda spatial_dims, in_channels, out_channels, kernel_size=3, stride=1, bias=False: \
        ref_shim.Convolution(spatial_dims, in_channels, out_channels, strides=stride, kernel_size=kernel_size, bias=bias,
                             conv_only=True)

    def get_upsample_layer(spatial_dims, in_channels, upsample_mode="nontrainable", scale_factor=2):
        assert str(getattr(upsample_mode, "value", upsample_mode)) == "nontrainable"
        sf = tuple(float(s) for s in scale_factor) if isinstance(scale_factor, (tuple, list)) else float(scale_factor)
        return nn.Upsample(scale_factor=sf, mode={2: "bilinear", 3: "trilinear"}[spatial_dims], align_corners=False)

    sb.get_upsample_layer = get_upsample_layer
    import monai.networks.layers.utils as lu

    def get_norm_layer(name, spatial_dims=1, channels=1):
        kind, kw = (name, {}) if isinstance(name, str) else (name[0], dict(name[1]))
        assert kind.lower() == "group"
        return nn.GroupNorm(num_channels=channels, **kw)

    def get_act_layer(name):
        kind, kw = (name, {}) if isinstance(name, str) else (name[0], dict(name[1]))
        return {"relu": nn.ReLU, "silu": nn.SiLU}[kind.lower()](**kw)

    lu.get_norm_layer, lu.get_act_layer = get_norm_layer, get_act_layer
    from nnunetv2.nets.seg_mamba import mamba_simple as ms
    ms.causal_conv1d_fn = None
    ms.selective_scan_fn = ref
    import mamba_ssm
    mamba_ssm.Mamba = lambda d_model, **kw: ms.Mamba(d_model, bimamba_type="none", use_fast_path=False, **kw)
    import nnunetv2.nets.lm2net as R
    return R


def main():
    R = bind()
    man = {}
    for name in ("LM2NetP", "LM2Net"):
        torch.manual_seed(0)
        net = getattr(R, name)(spatial_dims=2, in_ch=1, out_ch=2, deep_supervision=True, input_patch_size=(64, 64))
        man[name] = [[k, list(v.shape)] for k, v in net.state_dict().items()]
        det_fill(net)
        with torch.no_grad():  # keep A = -exp(A_log) in a sane range (the fill is centred on 0)
            for n, p in net.named_parameters():
                if n.endswith("A_log"):
                    p.copy_(torch.log(1.0 + torch.arange(p.numel(), dtype=torch.float32).reshape(p.shape) % 16) * 0.9 + 0.05 * p)
        # eval mode with unit BatchNorm running statistics: in train mode the RSU4F stages normalise 4 x 4 ... 8 x 8 maps by
        # the statistics of 32 ... 128 values, which amplifies fp32 rounding differences ~100x per stage and makes an
        # end-to-end comparison meaningless (first version of this fixture)
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.zero_()
                m.running_var.fill_(1.0)
        net.eval()
        x = torch.randn(1, 1, 64, 64, generator=torch.Generator().manual_seed(31))
        xg = x.clone().requires_grad_(True)
        # 64 strided samples + the rms of the output of every top-level child and of stage1's children (localises a
        # disagreement to a stage)
        mids, hooks = {}, []

        def tap(tag):
            def fn(mod, inp, out):
                o = out.detach().reshape(-1)
                mids[f"mid_{tag}"] = np.concatenate([o[::max(1, o.numel() // 64)][:64].numpy(),
                                                     [float(o.double().pow(2).mean().sqrt())]])
            return fn

        for tag, mod in list(net.named_children()) + [(f"stage1.{n}", m) for n, m in net.stage1.named_children()] + \
                [(f"stage1.down_layers.0.1.{n}", m) for n, m in net.stage1.down_layers[0][1].named_children()]:
            hooks.append(mod.register_forward_hook(tap(tag)))
        # FULL input / output of the modules behind stage 1: with the deterministic parameter fill every stage amplifies a
        # difference at its input ~10x (LayerNorm of the patch expansions over nearly constant tokens up to 50x), so only
        # stage 1 - whose input is x itself - can be compared end to end at fp32 precision; every later module is checked on
        # the reference's own input.  (Not stored: stage2d / level 1 - 1-4 MB tensors of module classes already covered.)
        full = {}

        def tap_full(tag):
            def fn(mod, inp, out):
                full[f"in_{tag}"] = inp[0].detach().numpy().copy()
                full[f"io_{tag}"] = out.detach().numpy().copy()
            return fn

        deep = [f"patch_merging{i}" for i in (1, 2, 3, 4)] + ["stage2", "stage3", "stage4", "stage5", "pool56", "stage6",
                                                               "stage5d"] + \
            [f"{k}{l}d" for l in (4, 3) for k in ("patch_expand", "concat_back_dim", "stage")] + \
            ["patch_expand2d", "concat_back_dim2d"]
        if name == "LM2Net":   # the wide net: the encoder-side tensors are 2-4x larger; the same classes are covered by LM2NetP
            deep = [t for t in deep if not t.startswith("patch_merging") and t not in ("stage2", "stage3")]
        for tag in deep:
            if hasattr(net, tag):
                hooks.append(getattr(net, tag).register_forward_hook(tap_full(tag)))
        outs = list(net(xg))
        for h in hooks:
            h.remove()
        np.savez_compressed(os.path.join(OUT, f"net_{name}_64.npz"), x=x.numpy(), **mids,
                            **{f"out{i}": o.detach().numpy() for i, o in enumerate(outs)})
        np.savez_compressed(os.path.join(OUT, f"netdeep_{name}_64.npz"), **{k: v.astype(np.float32) for k, v in full.items()})
        loss = 0
        for i, o in enumerate(outs):
            j = torch.arange(o.numel(), dtype=torch.float64)
            loss = loss + (o * torch.sin(0.37 * j + i).float().view_as(o)).sum() / o[0, 0].numel()
        loss.backward()
        # only the NAMES of the parameters that received a gradient are stored: with this parameter fill the gradient values
        # are not reproducible to better than 5-20 % between two fp32 implementations (tests/test_lm2net.py)
        gd = {}
        names = [n for n, p in net.named_parameters() if p.grad is not None]
        np.savez_compressed(os.path.join(OUT, f"netgrad_{name}_64.npz"), names=np.array(names), **gd)
        print(name, "params", sum(p.numel() for p in net.parameters()), "outs", [tuple(o.shape) for o in outs],
              "finite", all(bool(torch.isfinite(o).all()) for o in outs), flush=True)
    with open(os.path.join(OUT, "lm2net_manifest.json"), "w") as f:
        json.dump(man, f)
    light_munet()


def light_munet():
    """nets/LightMUNet.py LightMUNet (the stand-alone net of nnUNetTrainerLightMUNet), 2-D and 3-D, the trainer's
    configuration (init_filters 32, blocks (1, 2, 2, 4) / (1, 1, 1)): forward, dx, the names of the parameters that receive a gradient, state_dict manifest"""
    import nnunetv2.nets.LightMUNet as R
    man = {}
    for tag, sd, shape in (("2d", 2, (1, 1, 64, 64)), ("3d", 3, (1, 2, 16, 16, 16))):
        torch.manual_seed(0)
        net = R.LightMUNet(spatial_dims=sd, init_filters=32, in_channels=shape[1], out_channels=3,
                           blocks_down=[1, 2, 2, 4], blocks_up=[1, 1, 1])
        man[tag] = [[k, list(v.shape)] for k, v in net.state_dict().items()]
        det_fill(net)
        with torch.no_grad():
            for n, p in net.named_parameters():
                if n.endswith("A_log"):
                    p.copy_(torch.log(1.0 + torch.arange(p.numel(), dtype=torch.float32).reshape(p.shape) % 16) * 0.9 + 0.05 * p)
        net.train()
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(7))
        xg = x.clone().requires_grad_(True)
        y = net(xg)
        j = torch.arange(y.numel(), dtype=torch.float64)
        ((y * torch.sin(0.37 * j).float().view_as(y)).sum() / y[0, 0].numel()).backward()
        # gradient VALUES are not stored beyond dx (used for its norm): tools/probes/lightmunet_reference_conditioning.py
        gd = {"x": x.numpy(), "y": y.detach().numpy(), "dx": xg.grad.numpy()}
        names = [n for n, p in net.named_parameters() if p.grad is not None]
        np.savez_compressed(os.path.join(OUT, f"net_LightMUNet_{tag}.npz"), names=np.array(names), **gd)
        print("LightMUNet", tag, "params", sum(p.numel() for p in net.parameters()), "out", tuple(y.shape),
              "finite", bool(torch.isfinite(y).all()), flush=True)
    with open(os.path.join(OUT, "lightmunet_manifest.json"), "w") as f:
        json.dump(man, f)


def light_ss2d():
    """nets/LightSS2DMambaUNet.py LightSS2DMambaUNet (round 4), the trainer's configuration (2-D, init_filters 8, blocks (1, 2, 2,
    4) / (1, 1, 1)) built by the file's factory under torch.manual_seed(0), selective_scan_fn bound to the reference's
    selective_scan_ref: state_dict manifest + digest of the seeded parameters, forward, dx, the L2 norm of every parameter
    gradient and the reference's own response to a 1e-6 input perturbation (forward, dx, gradient norms)."""
    import nnunetv2.nets.LightSS2DMambaUNet as R
    R.selective_scan_fn = ref_shim.load_selective_scan_ref()
    import hashlib
    torch.manual_seed(0)
    net = R.get_mamband2net_from_plans(2, 1, 3)        # seeded construction + InitWeights_He: the parameters of the fixture
    man = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    h = hashlib.sha256()
    for v in net.state_dict().values():
        h.update(v.detach().contiguous().numpy().tobytes())
    digest = h.hexdigest()
    net.train()
    x = torch.randn(1, 1, 64, 64, generator=torch.Generator().manual_seed(7))

    def run(xin):
        net.zero_grad(set_to_none=True)
        xg = xin.clone().requires_grad_(True)
        y = net(xg)
        j = torch.arange(y.numel(), dtype=torch.float64)
        ((y * torch.sin(0.37 * j).float().view_as(y)).sum() / y[0, 0].numel()).backward()
        names = [n for n, p in net.named_parameters() if p.grad is not None]
        norms = np.array([float(p.grad.double().pow(2).sum().sqrt()) for n, p in net.named_parameters() if p.grad is not None])
        return y.detach(), xg.grad.clone(), names, norms

    y, dx, names, norms = run(x)
    i = torch.arange(x.numel(), dtype=torch.float64)
    y2, dx2, _, norms2 = run(x + 1e-6 * float(x.std()) * torch.cos(1.3 * i + 0.2).float().view_as(x))
    sens = {"y": float((y2 - y).abs().max() / y.abs().max()), "dx": float((dx2 - dx).abs().max() / dx.abs().max()),
            "gnorm_median": float(np.median(np.abs(norms2 - norms) / (norms + 1e-12)))}
    np.savez_compressed(os.path.join(OUT, "net_LightSS2DMambaUNet_2d.npz"), x=x.numpy(), y=y.numpy(), dx=dx.numpy(),
                        names=np.array(names), grad_norms=norms, sens=np.array([sens["y"], sens["dx"], sens["gnorm_median"]]))
    with open(os.path.join(OUT, "lightss2d_manifest.json"), "w") as f:
        json.dump({"state_dict": man, "seeded_sha256": digest}, f)
    print("LightSS2DMambaUNet params", sum(p.numel() for p in net.parameters()), "out", tuple(y.shape), "sens", sens, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "lightss2d":
        bind()
        light_ss2d()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lightmunet":
        bind()
        light_munet()
    else:
        main()
