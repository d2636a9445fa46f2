"""Generates tests/golden/*.npz from the REFERENCE's own code (imported from /root/reference under tools/ref_shim.py).
Run in the build container only:  python tools/make_golden.py
The fixtures are data (inputs, deterministic parameter fills, outputs, gradients); no reference source is copied."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def det_fill(module, skip=("A_logs", "Ds", "relative_position_index")):
    """Deterministic, RNG-free parameter fill shared with the tests (tests/golden_util.py restates it):
    k-th parameter (named_parameters order) <- scale_k * cos(0.7071 * i + k), i = flat index."""
    with torch.no_grad():
        for k, (name, p) in enumerate(module.named_parameters()):
            if name.split(".")[-1] in skip:
                continue
            n = p.numel()
            i = torch.arange(n, dtype=torch.float64)
            base = torch.cos(0.7071 * i + k).to(torch.float32).view_as(p)
            last = name.split(".")[-1]
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(base * (1.0 / fan_in ** 0.5))
            elif "norm" in name or last in ("weight",) and p.dim() == 1:
                p.copy_(1.0 + 0.1 * base if last == "weight" else 0.05 * base)
            else:
                p.copy_(0.05 * base)
        for k, (name, b) in enumerate(module.named_buffers()):
            if name.endswith("running_mean"):
                b.copy_(0.02 * torch.cos(torch.arange(b.numel(), dtype=torch.float32) + k))
            elif name.endswith("running_var"):
                b.copy_(1.0 + 0.1 * torch.cos(torch.arange(b.numel(), dtype=torch.float32) * 0.3 + k) ** 2)


def gen_selective_scan(ref):
    cases = {"a": (2, 4, 8, 16, 37), "b": (1, 4, 16, 16, 300), "c": (2, 6, 16, 16, 64), "d": (1, 4, 16, 16, 520)}
    for tag, (b, K, Dg, N, L) in cases.items():
        g = torch.Generator().manual_seed(ord(tag) + 7)
        KD = K * Dg
        u = torch.randn(b, KD, L, generator=g, requires_grad=True)
        delta = (torch.randn(b, KD, L, generator=g) * 0.5).requires_grad_(True)
        A = (-torch.exp(torch.log(torch.arange(1, N + 1, dtype=torch.float32))[None].repeat(KD, 1) +
                        0.1 * torch.randn(KD, N, generator=g))).requires_grad_(True)
        B = torch.randn(b, K, N, L, generator=g, requires_grad=True)
        C = torch.randn(b, K, N, L, generator=g, requires_grad=True)
        D = torch.randn(KD, generator=g, requires_grad=True)
        bias = (torch.randn(KD, generator=g) * 0.5 - 1.0).requires_grad_(True)
        y = ref(u, delta, A, B, C, D, None, bias, True)
        dy = torch.randn(y.shape, generator=g)
        grads = torch.autograd.grad(y, [u, delta, A, B, C, D, bias], dy)
        np.savez_compressed(os.path.join(OUT, f"selective_scan_{tag}.npz"),
                            u=u.detach().numpy(), delta=delta.detach().numpy(), A=A.detach().numpy(),
                            B=B.detach().numpy(), C=C.detach().numpy(), D=D.detach().numpy(),
                            delta_bias=bias.detach().numpy(), y=y.detach().numpy(), dy=dy.numpy(),
                            **{f"d{n}": gr.numpy() for n, gr in zip(["u", "delta", "A", "B", "C", "D", "bias"], grads)})
    # KAT-1 of SURVEY.md §8c
    u = torch.linspace(-1, 1, 8).view(1, 2, 4)
    delta = torch.linspace(.1, .8, 8).view(1, 2, 4)
    y = ref(u, delta, -torch.tensor([[1., 2.], [1., 2.]]), torch.linspace(.5, 1.5, 8).view(1, 1, 2, 4),
            torch.linspace(1, -1, 8).view(1, 1, 2, 4), torch.tensor([1., .5]), None, torch.tensor([0., -.5]), True)
    np.savez(os.path.join(OUT, "selective_scan_kat1.npz"), y=y.numpy())


def gen_losses():
    from nnunetv2.training.loss.compound_losses import DC_and_CE_loss
    from nnunetv2.training.loss.deep_supervision import DeepSupervisionWrapper
    from nnunetv2.training.loss.dice import MemoryEfficientSoftDiceLoss
    for tag, (shape, C, batch_dice) in {"3d": ((2, 9, 10, 11), 2, False), "2d": ((3, 24, 20), 4, True)}.items():
        g = torch.Generator().manual_seed(11)
        B, sp = shape[0], shape[1:]
        logits = (torch.randn(B, C, *sp, generator=g) * 2).requires_grad_(True)
        target = torch.randint(0, C, (B, 1, *sp), generator=g).to(torch.int16)
        loss = DC_and_CE_loss({'batch_dice': batch_dice, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {}, weight_ce=1,
                              weight_dice=1, ignore_label=None, dice_class=MemoryEfficientSoftDiceLoss)
        l = loss(logits, target)
        (gl,) = torch.autograd.grad(l, logits)
        # deep supervision on strided copies
        outs = [logits.detach(), logits.detach()[(..., *([slice(None, None, 2)] * len(sp)))].contiguous()]
        tgs = [target, target[(..., *([slice(None, None, 2)] * len(sp)))].contiguous()]
        outs.append(outs[1][(..., *([slice(None, None, 2)] * len(sp)))].contiguous())
        tgs.append(tgs[1][(..., *([slice(None, None, 2)] * len(sp)))].contiguous())
        w = np.array([1, 0.5, 0.0])
        w = w / w.sum()
        lds = DeepSupervisionWrapper(loss, w)(outs, tgs)
        np.savez_compressed(os.path.join(OUT, f"loss_{tag}.npz"), logits=logits.detach().numpy(), target=target.numpy(),
                            loss=l.detach().numpy(), dlogits=gl.numpy(), ds_loss=lds.detach().numpy(), ds_weights=w,
                            batch_dice=np.array(batch_dice))
        # ignore label (= C, the reference's convention: highest label + 1) on ~20 % of the voxels, and an all-ignored case
        tgt_ig = target.clone()
        tgt_ig[torch.rand(target.shape, generator=g) < 0.2] = C
        loss_ig = DC_and_CE_loss({'batch_dice': batch_dice, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {},
                                 weight_ce=1, weight_dice=1, ignore_label=C, dice_class=MemoryEfficientSoftDiceLoss)
        logits2 = logits.detach().clone().requires_grad_(True)
        l_ig = loss_ig(logits2, tgt_ig)
        (g_ig,) = torch.autograd.grad(l_ig, logits2)
        l_all = loss_ig(logits.detach(), torch.full_like(target, C))
        np.savez_compressed(os.path.join(OUT, f"loss_ignore_{tag}.npz"), target=tgt_ig.numpy(), loss=l_ig.detach().numpy(),
                            dlogits=g_ig.numpy(), loss_all_ignored=l_all.detach().numpy(), ignore_label=np.array(C))


def gen_tp_fp_fn():
    """Online-Dice statistics of validation_step: the reference's own get_tp_fp_fn_tn (training/loss/dice.py:122-180,
    imported as-is) driven exactly as nnUNetTrainer.validation_step does (nnUNetTrainer.py:1188-1216): argmax -> one-hot
    scatter (label maps, optional ignore label -> mask, target[ignore] = 0) or sigmoid > 0.5 (regions, optional ignore
    channel), axes = [0] + spatial.  Logits contain exact ties (argmax takes the first maximum)."""
    from nnunetv2.training.loss.dice import get_tp_fp_fn_tn
    out = {}
    for tag, (shape, C) in {"3d": ((2, 7, 9, 10), 3), "2d": ((3, 17, 23), 5)}.items():
        g = torch.Generator().manual_seed(23)
        B, sp = shape[0], shape[1:]
        logits = torch.round(torch.randn(B, C, *sp, generator=g) * 2) / 2          # coarse values: many exact ties
        target = torch.randint(0, C, (B, 1, *sp), generator=g).to(torch.int16)
        axes = [0] + list(range(2, logits.ndim))

        def label_stats(tgt, ignore_label):
            output_seg = logits.argmax(1)[:, None]
            onehot = torch.zeros(logits.shape, dtype=torch.float32)
            onehot.scatter_(1, output_seg, 1)
            tgt = tgt.clone()
            mask = None
            if ignore_label is not None:
                mask = (tgt != ignore_label).float()
                tgt[tgt == ignore_label] = 0
            tp, fp, fn, _ = get_tp_fp_fn_tn(onehot, tgt, axes=axes, mask=mask)
            return np.stack([tp.numpy(), fp.numpy(), fn.numpy()])

        out[f"{tag}_logits"], out[f"{tag}_target"] = logits.numpy(), target.numpy()
        out[f"{tag}_plain"] = label_stats(target, None)
        tgt_ig = target.clone()
        tgt_ig[torch.rand(target.shape, generator=g) < 0.25] = C
        out[f"{tag}_target_ignore"], out[f"{tag}_ignore_label"] = tgt_ig.numpy(), np.array(C)
        out[f"{tag}_ignore"] = label_stats(tgt_ig, C)
        # regions
        regions = (torch.rand(B, C, *sp, generator=g) < 0.4).to(torch.int16)
        ign = (torch.rand(B, 1, *sp, generator=g) < 0.2).to(torch.int16)
        pred = (torch.sigmoid(logits) > 0.5).long()
        tp, fp, fn, _ = get_tp_fp_fn_tn(pred, regions.bool(), axes=axes, mask=None)
        out[f"{tag}_regions"], out[f"{tag}_regions_ignore_channel"] = regions.numpy(), ign.numpy()
        out[f"{tag}_regions_plain"] = np.stack([tp.numpy(), fp.numpy(), fn.numpy()])
        tgt = torch.cat([regions, ign], 1)
        mask = 1 - tgt[:, -1:]
        tp, fp, fn, _ = get_tp_fp_fn_tn(pred, tgt[:, :-1].bool(), axes=axes, mask=mask)
        out[f"{tag}_regions_masked"] = np.stack([tp.numpy(), fp.numpy(), fn.numpy()])
    np.savez_compressed(os.path.join(OUT, "tp_fp_fn.npz"), **out)


def gen_region_losses():
    """DC_and_BCE_loss (region-based training) from the reference's module: plain and with the ignore-mask channel"""
    from nnunetv2.training.loss.compound_losses import DC_and_BCE_loss
    from nnunetv2.training.loss.dice import MemoryEfficientSoftDiceLoss
    for tag, (shape, C, batch_dice) in {"3d": ((2, 9, 10, 11), 3, False), "2d": ((3, 24, 20), 4, True)}.items():
        g = torch.Generator().manual_seed(17)
        B, sp = shape[0], shape[1:]
        logits = (torch.randn(B, C, *sp, generator=g) * 2)
        regions = (torch.rand(B, C, *sp, generator=g) < 0.35).to(torch.int16)
        ign = (torch.rand(B, 1, *sp, generator=g) < 0.2).to(torch.int16)
        out = dict(logits=logits.numpy(), regions=regions.numpy(), ignore=ign.numpy(), batch_dice=np.array(batch_dice))
        for name, use_ig, tgt in (("plain", False, regions), ("masked", True, torch.cat([regions, ign], 1))):
            loss = DC_and_BCE_loss({}, {'batch_dice': batch_dice, 'do_bg': True, 'smooth': 1e-5, 'ddp': False},
                                   use_ignore_label=use_ig, dice_class=MemoryEfficientSoftDiceLoss)
            x = logits.clone().requires_grad_(True)
            l = loss(x, tgt)
            (gr,) = torch.autograd.grad(l, x)
            out[f"{name}_loss"] = l.detach().numpy()
            out[f"{name}_dlogits"] = gr.numpy()
        np.savez_compressed(os.path.join(OUT, f"loss_regions_{tag}.npz"), **out)


def gen_window_attention():
    from nnunetv2.nets import swt2net
    for tag, (dim, heads, shift, HW) in {"s": (32, 2, True, 14), "n": (64, 4, False, 21)}.items():
        m = swt2net.WindowAttention(dim=dim, window_size=7, num_heads=heads, shift=shift)
        det_fill(m)
        m.eval()
        g = torch.Generator().manual_seed(3)
        x = torch.randn(2, HW, HW, dim, generator=g, requires_grad=True)
        y = m(x)
        dy = torch.randn(y.shape, generator=g)
        params = [p for _, p in m.named_parameters()]
        grads = torch.autograd.grad(y, [x] + params, dy)
        np.savez_compressed(os.path.join(OUT, f"window_attention_{tag}.npz"), x=x.detach().numpy(), y=y.detach().numpy(),
                            dy=dy.numpy(), dx=grads[0].numpy(), cfg=np.array([dim, heads, int(shift), HW]),
                            **{"g_" + n: gr.numpy() for (n, _), gr in zip(m.named_parameters(), grads[1:])})
    # KAT-4 of SURVEY.md §8c
    m = swt2net.WindowAttention(dim=4, window_size=7, num_heads=2, shift=True).eval()
    with torch.no_grad():
        for _, p in m.named_parameters():
            p.copy_(torch.linspace(-.5, .5, p.numel()).view_as(p))
    out = m(torch.linspace(-1, 1, 784).view(1, 14, 14, 4))
    np.savez(os.path.join(OUT, "window_attention_kat4.npz"), out=out.detach().numpy())
    # a whole SwinTransformerBlock incl. the top/left padding quirk (19 is not a multiple of 7)
    blk = swt2net.SwinTransformerBlock(dim=32, num_heads=2, window_size=7, shift=True, drop_path=0.0) \
        if "shift" in swt2net.SwinTransformerBlock.__init__.__code__.co_varnames else None
    if blk is not None:
        det_fill(blk)
        blk.eval()
        x = torch.randn(1, 19, 19, 32, generator=torch.Generator().manual_seed(5))
        np.savez_compressed(os.path.join(OUT, "swin_block.npz"), x=x.numpy(), y=blk(x).detach().numpy())


def gen_ss2d():
    from nnunetv2.nets import m2net
    m = m2net.SS2D(d_model=16)
    det_fill(m)
    m.eval()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 12, 10, 16, generator=g, requires_grad=True)
    y = m(x)
    dy = torch.randn(y.shape, generator=g)
    params = list(m.named_parameters())
    grads = torch.autograd.grad(y, [x] + [p for _, p in params], dy)
    np.savez_compressed(os.path.join(OUT, "ss2d.npz"), x=x.detach().numpy(), y=y.detach().numpy(), dy=dy.numpy(),
                        dx=grads[0].numpy(), names=np.array([n for n, _ in params]),
                        **{"g_" + n: gr.numpy() for (n, _), gr in zip(params, grads[1:])})


def gen_ssnd():
    from nnunetv2.nets import ssnd2net
    for tag, (sd, shp) in {"3d": (3, (1, 4, 6, 8, 8)), "2d": (2, (2, 6, 10, 8))}.items():
        m = ssnd2net.SSND(spatial_dims=sd, factorization_type="cross-scan", d_model=shp[-1])
        det_fill(m)
        m.eval()
        g = torch.Generator().manual_seed(13)
        x = torch.randn(*shp, generator=g, requires_grad=True)
        y = m(x)
        dy = torch.randn(y.shape, generator=g)
        params = list(m.named_parameters())
        grads = torch.autograd.grad(y, [x] + [p for _, p in params], dy)
        np.savez_compressed(os.path.join(OUT, f"ssnd{tag}.npz"), x=x.detach().numpy(), y=y.detach().numpy(),
                            dy=dy.numpy(), dx=grads[0].numpy(), names=np.array([n for n, _ in params]),
                            **{"g_" + n: gr.numpy() for (n, _), gr in zip(params, grads[1:])})


def gen_nets():
    from nnunetv2.nets import m2net, swt2net
    man = {}
    for name, ctor, size in [("M2NetP", lambda: m2net.M2NetP(1, 2, True), 64), ("SwT2Net", lambda: swt2net.SwT2Net(1, 2, True), 64)]:
        torch.manual_seed(0)
        net = ctor()
        det_fill(net)
        net.eval()
        x = torch.randn(1, 1, size, size, generator=torch.Generator().manual_seed(21))
        with torch.no_grad():
            outs = net(x)
        np.savez_compressed(os.path.join(OUT, f"net_{name}_{size}.npz"), x=x.numpy(),
                            **{f"out{i}": o.numpy() for i, o in enumerate(outs)})
        man[name] = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        # whole-net BACKWARD (eval mode: running BatchNorm statistics, no DropPath): loss = sum_i <out_i, G_i> with
        # formula-made G_i; dx in full, of every parameter gradient <= 256 evenly strided samples + its L2 norm
        xg = x.clone().requires_grad_(True)
        outs = net(xg)
        loss = 0
        for i, o in enumerate(outs):
            j = torch.arange(o.numel(), dtype=torch.float64)
            loss = loss + (o * torch.sin(0.37 * j + i).float().view_as(o)).sum() / o[0, 0].numel()
        loss.backward()
        gd = {"dx": xg.grad.numpy()}
        names = []
        for k, (n, p) in enumerate(net.named_parameters()):
            if p.grad is None:
                continue
            g = p.grad.reshape(-1)
            names.append(n)
            gd[f"g{k}"] = g[::max(1, g.numel() // 256)][:256].numpy()
            gd[f"n{k}"] = np.array(float(g.double().norm()))
        np.savez_compressed(os.path.join(OUT, f"netgrad_{name}_{size}.npz"), names=np.array(names), **gd)
    # the benchmark model itself (M2Net, not the small variant): forward fixture
    torch.manual_seed(0)
    net = m2net.M2Net(1, 2, True)
    det_fill(net)
    net.eval()
    x = torch.randn(1, 1, 64, 64, generator=torch.Generator().manual_seed(22))
    with torch.no_grad():
        outs = net(x)
    np.savez_compressed(os.path.join(OUT, "net_M2Net_64.npz"), x=x.numpy(), **{f"out{i}": o.numpy() for i, o in enumerate(outs)})
    torch.manual_seed(0)
    man["M2Net"] = [(k, tuple(v.shape)) for k, v in m2net.M2Net(1, 2, True).state_dict().items()]
    import json
    with open(os.path.join(OUT, "state_dict_manifest.json"), "w") as f:
        json.dump({k: [[n, list(s)] for n, s in v] for k, v in man.items()}, f)


def gen_sliding_window():
    """Reference tile geometry, importance maps and the reference's own accumulation loop on CPU half tensors
    (nnUNetPredictor._internal_predict_sliding_window_return_logits with do_on_device=False) around a toy network
    whose outputs are bit-reproducible (tests/golden_util.toy_seg_network)."""
    import types
    from nnunetv2.inference.sliding_window_prediction import compute_gaussian, compute_steps_for_sliding_window
    from nnunetv2.inference.predict_from_raw_data import nnUNetPredictor
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    from golden_util import toy_image, toy_seg_network
    out = {}
    step_cases = [((110,), (64,), 0.5), ((128, 128, 128), (128, 128, 128), 0.5), ((300, 257, 140), (128, 128, 96), 0.5),
                  ((77, 512), (64, 128), 0.25), ((65, 64, 200), (64, 64, 64), 1.0), ((512, 512), (512, 512), 0.5),
                  ((193, 130), (64, 96), 0.75)]
    for i, (img, tile, st) in enumerate(step_cases):
        steps = compute_steps_for_sliding_window(img, tile, st)
        out[f"steps{i}_args"] = np.array(list(img) + list(tile) + [st], dtype=np.float64)
        for a, s_ in enumerate(steps):
            out[f"steps{i}_axis{a}"] = np.array(s_, dtype=np.int64)
    out["n_step_cases"] = np.array(len(step_cases))
    for i, tile in enumerate([(16, 24, 20), (32, 48), (128, 128, 128)]):
        g = compute_gaussian(tuple(tile), sigma_scale=1. / 8, value_scaling_factor=10, device=torch.device("cpu"))
        if i < 2:
            out[f"gauss{i}"] = g.numpy()
        else:  # 4 MB as fp16: keep a strided sample + checksums
            out[f"gauss{i}_sample"] = g[::9, ::7, ::5].numpy()
            out[f"gauss{i}_sum_min_max"] = np.array([float(g.double().sum()), float(g.min()), float(g.max())])
        out[f"gauss{i}_tile"] = np.array(tile)

    class _Net(torch.nn.Module):
        def forward(self, x):
            return toy_seg_network(x)

    run_cases = [
        # image (c, *dims), patch, step, gaussian, mirroring axes
        ("3d_mirror", (20, 29, 27), (16, 16, 16), 0.5, True, (0, 1, 2)),
        ("3d_plain", (17, 16, 40), (16, 16, 16), 0.5, False, None),
        ("2d_on_3d", (3, 40, 37), (32, 24), 0.5, True, (0, 1)),
        ("3d_mirror2", (24, 16, 16), (16, 16, 16), 0.25, True, (0, 2)),
    ]
    for name, dims, patch, st, gauss, mirror in run_cases:
        pr = nnUNetPredictor(tile_step_size=st, use_gaussian=gauss, use_mirroring=mirror is not None,
                             perform_everything_on_device=False, device=torch.device("cpu"), verbose=False,
                             allow_tqdm=False)
        pr.network = _Net()
        pr.configuration_manager = types.SimpleNamespace(patch_size=list(patch))
        pr.label_manager = types.SimpleNamespace(num_segmentation_heads=2)
        pr.allowed_mirroring_axes = mirror
        data = toy_image(dims, seed=len(name))
        slicers = pr._internal_get_sliding_window_slicers(data.shape[1:])
        with torch.no_grad():
            logits = pr._internal_predict_sliding_window_return_logits(data, slicers, False)
        assert logits.dtype == torch.float16
        out[f"run_{name}_logits"] = logits.numpy()
        out[f"run_{name}_cfg"] = np.array(list(dims) + list(patch) + [st, float(gauss), len(name)], dtype=np.float64)
        out[f"run_{name}_mirror"] = np.array(mirror if mirror is not None else [], dtype=np.int64)
        out[f"run_{name}_nslicers"] = np.array(len(slicers))
    np.savez_compressed(os.path.join(OUT, "sliding_window.npz"), **out)


def gen_ssnd2net():
    """MU-stage forward fixtures (2-D 96^2, 3-D 24^3) + state_dict manifests of SSND2Net / SSND2NetP in both
    dimensionalities, from the reference's nets/ssnd2net.py under the shim."""
    import json
    from nnunetv2.nets import ssnd2net
    man = {}
    # One MU stage per dimensionality (encoder + decoder of VSS/SSND blocks, patch merge / expand incl. axes that stop
    # pooling at odd extents).  Whole-network outputs are NOT used as fixtures: with ~100 LayerNorm / InstanceNorm
    # layers in sequence a 1e-6 difference between two correct scan implementations grows to 20 % of the output
    # (measured with selective_scan_ref vs the oracle scan on the same weights); the outer wiring is pinned by the
    # state_dict manifests below and by tools/check_ssnd2net_wiring.py (bit-identical outputs when both nets share one
    # scan implementation).
    for tag, patch, cin, nl in [("2d", (96, 96), 1, 7), ("3d", (24, 24, 24), 8, 5)]:
        torch.manual_seed(0)
        mu = ssnd2net.MU(spatial_dims=len(patch), factorization_type="cross-scan", in_ch=cin, mid_ch=16, out_ch=64,
                         n_layers=nl, input_patch_size=tuple(patch), patch_size=1, add_last=True)
        det_fill(mu)
        mu.eval()
        i = torch.arange(cin * int(np.prod(patch)), dtype=torch.float64)
        x = torch.cos(0.173 * i + 0.3).float().reshape(1, cin, *patch)
        with torch.no_grad():
            y = mu(x)
        # x is regenerated from its formula by the test; of y keep every 8th channel + per-channel moments of all
        np.savez_compressed(os.path.join(OUT, f"ssnd2net_MU_{tag}.npz"), y_sub=y[:, ::8].numpy(),
                            y_mean=y.double().mean(dim=tuple(range(2, y.dim()))).numpy(),
                            y_absmean=y.double().abs().mean(dim=tuple(range(2, y.dim()))).numpy(),
                            cfg=np.array([cin, 16, 64, nl]))
    for tag, patch in [("2d", (96, 96)), ("3d", (24, 24, 24))]:
        torch.manual_seed(0)
        net = ssnd2net.SSND2NetP(spatial_dims=len(patch), factorization_type="cross-scan", in_ch=1, out_ch=2,
                                 deep_supervision=True, input_patch_size=list(patch))
        man[f"SSND2NetP_{tag}"] = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        torch.manual_seed(0)
        big = ssnd2net.SSND2Net(spatial_dims=len(patch), factorization_type="cross-scan", in_ch=1, out_ch=2,
                                deep_supervision=True, input_patch_size=list(patch))
        man[f"SSND2Net_{tag}"] = [(k, tuple(v.shape)) for k, v in big.state_dict().items()]
    import gzip
    with gzip.open(os.path.join(OUT, "state_dict_manifest_ssnd2net.json.gz"), "wt") as f:
        json.dump({k: [[n, list(s_)] for n, s_ in v] for k, v in man.items()}, f)


def gen_mamba(ref):
    """1-D Mamba block: the reference's vendored module (nets/seg_mamba/mamba_simple.py) run on CPU.
    bimamba "none" goes through its slow path (conv1d + act, x/dt projections, selective_scan_ref with z);
    "v2" / "v3" go through its own fast-path composition code with mamba_inner_fn_no_out_proj bound to the reference's
    mamba_inner_ref (identity out_proj), see tools/ref_shim.load_mamba_inner_ref."""
    from nnunetv2.nets.seg_mamba import mamba_simple as ms
    inner_ref = ref_shim.load_mamba_inner_ref()
    ms.causal_conv1d_fn = None
    ms.selective_scan_fn = ref

    def no_out_proj(xz, cw, cb, xw, dw, A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                    C_proj_bias=None, delta_softplus=True):
        eye = torch.eye(A.shape[0], dtype=xz.dtype)
        return inner_ref(xz, cw, cb, xw, dw, eye, None, A, B, C, D, delta_bias, B_proj_bias, C_proj_bias,
                         delta_softplus).transpose(1, 2)

    ms.mamba_inner_fn_no_out_proj = no_out_proj
    out = {}
    for tag, kind, fast, d_model, L, ns in [("none", "none", False, 48, 37, 5), ("v2", "v2", True, 32, 24, 5),
                                            ("v3", "v3", True, 32, 40, 4)]:
        torch.manual_seed(0)
        m = ms.Mamba(d_model, bimamba_type=kind, nslices=ns, use_fast_path=fast)
        det_fill(m, skip=())
        with torch.no_grad():  # keep A = -exp(A_log) and softplus(dt bias) in a sane range
            for n, p in m.named_parameters():
                if "A_" in n or n == "A_log":
                    p.copy_(torch.log(1.0 + torch.arange(p.numel(), dtype=torch.float32).reshape(p.shape) % 16) * 0.9 + 0.05 * p)
        Bn = 2
        i = torch.arange(Bn * L * d_model, dtype=torch.float64)
        x = torch.cos(0.37 * i + 0.5).float().reshape(Bn, L, d_model).requires_grad_(True)
        G = torch.sin(0.11 * i + 1.0).float().reshape(Bn, L, d_model)
        y = m(x)
        (y * G).sum().backward()
        out[f"{tag}_x"] = x.detach().numpy()
        out[f"{tag}_G"] = G.numpy()
        out[f"{tag}_y"] = y.detach().numpy()
        out[f"{tag}_dx"] = x.grad.numpy()
        out[f"{tag}_cfg"] = np.array([d_model, L, ns])
        for n, p in m.named_parameters():
            out[f"{tag}_p_{n}"] = p.detach().numpy()
            if p.grad is not None:
                out[f"{tag}_g_{n}"] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "mamba_block.npz"), **out)


def gen_mambandcore(ref):
    """MambaNDCore (nets/mamba_nd2net.py:725-1001: patch embedding + ordered / reversed Mamba blocks) of the reference,
    run on CPU with `mamba_ssm.Mamba` bound to the reference's own vendored block (nets/seg_mamba/mamba_simple.py, slow
    path on selective_scan_ref - same mathematics as mamba_ssm's block, bimamba "none"), 2-D and 3-D."""
    from nnunetv2.nets.seg_mamba import mamba_simple as ms
    ms.causal_conv1d_fn = None
    ms.selective_scan_fn = ref
    import nnunetv2.nets.mamba_nd2net as R
    R.Mamba = lambda dim, layer_idx=None, **kw: ms.Mamba(dim, bimamba_type="none", use_fast_path=False,
                                                         layer_idx=layer_idx, **kw)
    out = {}
    for tag, sd, img, patch, cin, E, nl in [("2d", 2, (32, 48), (4, 4), 3, 32, 5), ("3d", 3, (16, 16, 16), (4, 4, 4), 2, 48, 7)]:
        torch.manual_seed(0)
        core = R.MambaNDCore(spatial_dims=sd, in_channels=cin, img_size=img, patch_size=patch, embed_dims=E,
                             drop_path_rate=0.0, final_norm=False, fused_add_norm=False, drop_rate=0.0, d_state=16,
                             force_a2=False, pretrained=None, num_layers=nl).train()
        det_fill(core, skip=())
        with torch.no_grad():
            for n, p in core.named_parameters():
                if n.endswith("A_log") or "A_b_log" in n or "A_s_log" in n:
                    p.copy_(torch.log(1.0 + torch.arange(p.numel(), dtype=torch.float32).reshape(p.shape) % 16) * 0.9 + 0.05 * p)
        i = torch.arange(2 * cin * int(np.prod(img)), dtype=torch.float64)
        x = torch.cos(0.37 * i + 0.5).float().reshape(2, cin, *img).requires_grad_(True)
        y, outs = core(x)
        G = torch.sin(0.11 * torch.arange(y.numel(), dtype=torch.float64) + 1.0).float().reshape(y.shape)
        (y * G).sum().backward()
        out[f"{tag}_cfg"] = np.array([sd, cin, E, nl, *img, *patch])
        out[f"{tag}_x"], out[f"{tag}_G"], out[f"{tag}_y"] = x.detach().numpy(), G.numpy(), y.detach().numpy()
        out[f"{tag}_mid"] = outs[2].detach().numpy()
        out[f"{tag}_dx"] = x.grad.numpy()
        for n, p in core.named_parameters():
            if any(t in n for t in ("_b.", "_s.", "_b_log", "_s_log", ".D_b", ".D_s")):
                continue                       # tensors of the vendored class that mamba_ssm.Mamba does not have
            out[f"{tag}_p_{n}"] = p.detach().numpy()
            if p.grad is not None:
                out[f"{tag}_g_{n}"] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "mambandcore.npz"), **out)


def state_digest(sd):
    """sha256 over every tensor's bytes in state_dict order + crc32 of every 40th tensor (restated in the test)"""
    import hashlib
    import zlib
    h = hashlib.sha256()
    crc = {}
    for i, (k, v) in enumerate(sd.items()):
        b = v.detach().cpu().contiguous().numpy().tobytes()
        h.update(b)
        if i % 40 == 0:
            crc[k] = zlib.crc32(b)
    return {"sha256": h.hexdigest(), "n_tensors": len(sd), "crc32": crc}


def gen_seeded_init():
    """tests/golden/seeded_init.json: digests of the reference's networks constructed under torch.manual_seed(0) - the
    product's constructors must draw the same RNG stream (creation order, the RNG-advancing fake init of VSSLayer)."""
    import json
    from nnunetv2.nets import m2net, ssnd2net, swt2net
    out = {}
    for name, ctor in _seeded_models(m2net, swt2net, ssnd2net).items():
        torch.manual_seed(0)
        out[name] = state_digest(ctor().state_dict())
    with open(os.path.join(OUT, "seeded_init.json"), "w") as f:
        json.dump(out, f)


def _seeded_models(m2net, swt2net, ssnd2net):
    kw2 = dict(spatial_dims=2, factorization_type="cross-scan", in_ch=1, out_ch=2, deep_supervision=True, input_patch_size=[96, 96])
    kw3 = dict(spatial_dims=3, factorization_type="cross-scan", in_ch=1, out_ch=2, deep_supervision=True, input_patch_size=[24, 24, 24])
    return {"SS2D_16": lambda: m2net.SS2D(d_model=16), "M2NetP": lambda: m2net.M2NetP(1, 2, True),
            "M2Net": lambda: m2net.M2Net(1, 2, True), "SwT2Net": lambda: swt2net.SwT2Net(1, 2, True),
            "SSND2NetP_2d": lambda: ssnd2net.SSND2NetP(**kw2), "SSND2Net_3d": lambda: ssnd2net.SSND2Net(**kw3)}


def gen_dataloader_bbox():
    """tests/golden/dataloader_bbox.json: outputs of the reference's OWN nnUNetDataLoader.get_bbox
    (training/dataloading/data_loader.py:102-178).  The class cannot be imported (its base class lives in batchgenerators,
    absent), so the method's source is taken from the file by ast and executed unchanged against a plain namespace that
    carries the four attributes it reads (need_to_pad, patch_size, has_ignore, annotated_classes_key).  Scenarios are
    regenerated by the test from the stored parameters; the numpy seed makes the RNG stream part of the fixture."""
    import ast
    import json
    import types
    import warnings
    path = os.path.join(ref_shim.REF, "nnunetv2/training/dataloading/data_loader.py")
    src = open(path).read()
    tree = ast.parse(src)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "nnUNetDataLoader"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "get_bbox"][0]
    mod = ast.Module(body=[fn], type_ignores=[])
    import typing
    ns = {"np": np, "warnings": warnings, "Union": typing.Union, "Tuple": typing.Tuple}
    exec(compile(mod, path, "exec"), ns)
    get_bbox = ns["get_bbox"]
    out = []
    for sc in _bbox_scenarios():
        self_ = types.SimpleNamespace(need_to_pad=np.array(sc["need_to_pad"]), patch_size=tuple(sc["patch_size"]),
                                      has_ignore=sc["has_ignore"], annotated_classes_key=tuple(sc["annotated_classes_key"]))
        np.random.seed(sc["seed"])
        cl = _bbox_class_locations(sc)
        res = []
        for i in range(8):
            lbs, ubs = get_bbox(self_, np.array(sc["data_shape"]), sc["force_fg"][i % len(sc["force_fg"])], cl)
            res.append([[int(v) for v in lbs], [int(v) for v in ubs]])
        out.append(dict(sc, boxes=res))
    with open(os.path.join(OUT, "dataloader_bbox.json"), "w") as f:
        json.dump(out, f)


def _bbox_scenarios():
    return [
        dict(seed=1, data_shape=[40, 56, 48], patch_size=[32, 32, 32], need_to_pad=[8, 8, 8], has_ignore=False,
             annotated_classes_key=[-1, 0, 1, 2], force_fg=[False], classes="none"),
        dict(seed=2, data_shape=[40, 56, 48], patch_size=[32, 32, 32], need_to_pad=[8, 8, 8], has_ignore=False,
             annotated_classes_key=[-1, 0, 1, 2], force_fg=[False, True], classes="two"),
        dict(seed=3, data_shape=[20, 30, 64], patch_size=[32, 32, 32], need_to_pad=[0, 4, 9], has_ignore=False,
             annotated_classes_key=[-1, 0, 1], force_fg=[True], classes="two"),          # case smaller than the patch
        dict(seed=4, data_shape=[1, 50, 40], patch_size=[1, 32, 32], need_to_pad=[0, 6, 6], has_ignore=True,
             annotated_classes_key=[-1, 0, 1, 2], force_fg=[False, True], classes="ignore"),   # pseudo 3-D, ignore label
        dict(seed=5, data_shape=[48, 48, 48], patch_size=[32, 32, 32], need_to_pad=[8, 8, 8], has_ignore=False,
             annotated_classes_key=[-1, 0, 1, 2], force_fg=[True], classes="empty"),     # no foreground at all
    ]


def _bbox_class_locations(sc):
    """class_locations as the preprocessing stores them: per class an (n, 4) integer array (channel, z, y, x)"""
    if sc["classes"] == "none":
        return None
    rs = np.random.RandomState(100 + sc["seed"])
    sh = sc["data_shape"]

    def locs(n):
        return np.stack([np.zeros(n, dtype=np.int64)] + [rs.randint(0, s, n) for s in sh], 1)

    if sc["classes"] == "two":
        return {1: locs(40), 2: locs(7), 3: np.zeros((0, 4), dtype=np.int64)}
    if sc["classes"] == "ignore":
        return {1: locs(30), 2: np.zeros((0, 4), dtype=np.int64), tuple(sc["annotated_classes_key"]): locs(60)}
    return {1: np.zeros((0, 4), dtype=np.int64), 2: np.zeros((0, 4), dtype=np.int64)}


if __name__ == "__main__":
    ref = ref_shim.install()
    which = sys.argv[1:] or ["scan", "loss", "attn", "ss2d", "ssnd", "nets", "sw", "mamba", "ssnd2net", "mambandcore", "bbox", "seeded"]
    if "sw" in which:
        gen_sliding_window()
    if "mamba" in which:
        gen_mamba(ref)
    if "mambandcore" in which:
        gen_mambandcore(ref)
    if "ssnd2net" in which:
        gen_ssnd2net()
    if "scan" in which:
        gen_selective_scan(ref)
    if "loss" in which:
        gen_losses()
        gen_region_losses()
        gen_tp_fp_fn()
    if "attn" in which:
        gen_window_attention()
    if "ss2d" in which:
        gen_ss2d()
    if "ssnd" in which:
        gen_ssnd()
    if "nets" in which:
        gen_nets()
    if "bbox" in which:
        gen_dataloader_bbox()
    if "seeded" in which:
        gen_seeded_init()
    print(sorted(os.listdir(OUT)))
