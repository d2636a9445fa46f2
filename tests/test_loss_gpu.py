"""GPU parity of the fused Dice+CE HIP loss (through the C-ABI) against the CPU oracle (oracle/losses.py, which is
pinned to the reference by KAT-2/KAT-3 and tests/golden/loss_*.npz).  fp32 logits: rtol 1e-5 on the loss,
1e-4 on the gradient (the kernel uses __expf/__logf); fp16 logits: gradient compared after fp16 rounding."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import losses as O
from nnuzoo_amd.training.loss import DC_and_CE_loss, DeepSupervisionWrapper, MemoryEfficientSoftDiceLoss


def make(batch_dice):
    return DC_and_CE_loss({'batch_dice': batch_dice, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {}, weight_ce=1,
                          weight_dice=1, ignore_label=None, dice_class=MemoryEfficientSoftDiceLoss)


def test_kat2_kat3(hip_lib):
    logits = torch.stack([torch.linspace(-2, 2, 32).view(2, 4, 4), torch.linspace(1, -1, 32).view(2, 4, 4)], 1)
    target = (torch.arange(32).view(2, 1, 4, 4) % 3 == 0).to(torch.int16)
    l = make(True)(logits.cuda(), target.cuda())
    assert abs(float(l) - 0.5730621) < 2e-6
    outs = [logits, logits[..., ::2, ::2].contiguous(), logits[..., ::4, ::4].contiguous()]
    tg = [target, target[..., ::2, ::2].contiguous(), target[..., ::4, ::4].contiguous()]
    w = DeepSupervisionWrapper(make(True), [2 / 3, 1 / 3, 0])
    l3 = w([o.cuda() for o in outs], [t.cuda() for t in tg])
    assert abs(float(l3) - 0.6382819) < 2e-6


@pytest.mark.parametrize("shape,C,batch_dice", [((2, 17, 19, 23), 2, False), ((3, 64, 48), 4, True), ((2, 8, 8, 8), 3, False)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_loss_value_and_grad(hip_lib, shape, C, batch_dice, dtype):
    g = torch.Generator().manual_seed(3)
    B, sp = shape[0], shape[1:]
    logits = (torch.randn(B, C, *sp, generator=g) * 2).to(dtype)
    target = torch.randint(0, C, (B, 1, *sp), generator=g).to(torch.int16)
    ref_in = logits.float().requires_grad_(True)
    ref = O.dc_and_ce(ref_in, target, batch_dice)
    ref.backward()
    x = logits.cuda().detach().clone().requires_grad_(True)
    l = make(batch_dice)(x, target.cuda())
    l.backward()
    torch.cuda.synchronize()
    assert abs(l.item() - ref.item()) <= 2e-5 * max(1.0, abs(ref.item())), (l.item(), ref.item())
    got = x.grad.float().cpu()
    tol = 1e-4 if dtype == torch.float32 else 2e-3
    scale = ref_in.grad.abs().max().item()
    assert torch.allclose(got, ref_in.grad, rtol=tol, atol=tol * scale), (got - ref_in.grad).abs().max().item()


def test_cpu_tensor_raises():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        make(False)(torch.zeros(1, 2, 4, 4), torch.zeros(1, 1, 4, 4, dtype=torch.int16))


@pytest.mark.parametrize("shape,dtype", [((2, 3, 9, 17, 11), torch.float16), ((3, 2, 64, 50), torch.float32),
                                         ((1, 8, 5, 6, 7), torch.float16)])
def test_argmax_tp_fp_fn_matches_reference_formula(hip_lib, shape, dtype):
    """validation statistics kernel vs the reference's argmax -> scatter -> get_tp_fp_fn_tn chain (oracle), exact;
    quantised logits force ties (first maximum wins, like torch.argmax)"""
    from nnuzoo_amd import hip_ops as ops
    from oracle.losses import tp_fp_fn_hard
    g = torch.Generator().manual_seed(7)
    logits = (torch.randn(shape, generator=g) * 2).round().to(dtype)       # many exact ties
    target = torch.randint(0, shape[1], (shape[0], 1, *shape[2:]), generator=g).to(torch.int16)
    tp, fp, fn = ops.argmax_tp_fp_fn(logits.cuda(), target.cuda())
    rtp, rfp, rfn = tp_fp_fn_hard(logits.float(), target)
    assert torch.equal(tp.cpu().float(), rtp) and torch.equal(fp.cpu().float(), rfp) and torch.equal(fn.cpu().float(), rfn)
    assert int((tp + fn).sum()) == target.numel()


@pytest.mark.parametrize("tag", ["2d", "3d"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_validation_statistics_match_reference_golden(hip_lib, tag, dtype):
    """nnz_argmax_tp_fp_fn / nnz_region_tp_fp_fn against tests/golden/tp_fp_fn.npz: the outputs of the reference's own
    get_tp_fp_fn_tn (dice.py:122-180) driven as validation_step drives it (nnUNetTrainer.py:1188-1216) - label maps
    with exact ties, ignore label, regions with and without the ignore channel.  Exact counts (the golden logits are
    multiples of 0.5, so the fp16 copy is the same number)."""
    import os
    from nnuzoo_amd import hip_ops as ops
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "tp_fp_fn.npz"))
    logits = torch.from_numpy(z[f"{tag}_logits"]).to(dtype).cuda()
    def stack(t):
        return torch.stack([v.cpu().float() for v in t]).numpy()
    assert np.array_equal(stack(ops.argmax_tp_fp_fn(logits, torch.from_numpy(z[f"{tag}_target"]).cuda())),
                          z[f"{tag}_plain"])
    assert np.array_equal(stack(ops.argmax_tp_fp_fn(logits, torch.from_numpy(z[f"{tag}_target_ignore"]).cuda(),
                                                    int(z[f"{tag}_ignore_label"]))), z[f"{tag}_ignore"])
    r, ig = torch.from_numpy(z[f"{tag}_regions"]), torch.from_numpy(z[f"{tag}_regions_ignore_channel"])
    assert np.array_equal(stack(ops.region_tp_fp_fn(logits, r.cuda())), z[f"{tag}_regions_plain"])
    assert np.array_equal(stack(ops.region_tp_fp_fn(logits, torch.cat([r, ig], 1).cuda())), z[f"{tag}_regions_masked"])


@pytest.mark.parametrize("tag", ["2d", "3d"])
def test_ignore_label_matches_reference_golden(hip_lib, tag):
    """fused kernel with ignore_label against the reference's DC_and_CE_loss(ignore_label=C) fixtures: loss, gradient
    (zero on ignored voxels), the all-ignored batch, and the validation statistics with the same mask"""
    import os
    from nnuzoo_amd import hip_ops as ops
    gold = os.path.join(os.path.dirname(__file__), "golden")
    g, gi = np.load(os.path.join(gold, f"loss_{tag}.npz")), np.load(os.path.join(gold, f"loss_ignore_{tag}.npz"))
    ig, bd = int(gi["ignore_label"]), bool(g["batch_dice"])
    loss = DC_and_CE_loss({'batch_dice': bd, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {}, weight_ce=1,
                          weight_dice=1, ignore_label=ig, dice_class=MemoryEfficientSoftDiceLoss)
    x = torch.from_numpy(g["logits"]).cuda().requires_grad_(True)
    t = torch.from_numpy(gi["target"]).cuda()
    l = loss(x, t)
    l.backward()
    assert abs(float(l) - float(gi["loss"])) < 2e-5
    ref_g = torch.from_numpy(gi["dlogits"])
    assert torch.allclose(x.grad.cpu(), ref_g, rtol=1e-4, atol=1e-4 * ref_g.abs().max().item())
    assert x.grad.cpu()[(t.cpu() == ig).expand_as(x)].abs().max().item() == 0
    l_all = loss(x.detach(), torch.full_like(t, ig))
    assert abs(float(l_all) - float(gi["loss_all_ignored"])) < 1e-6
    # validation statistics: ignored voxels count nowhere (nnUNetTrainer.validation_step's mask)
    tp, fp, fn = ops.argmax_tp_fp_fn(x.detach(), t, ig)
    keep = (t != ig)
    tt = torch.where(keep, t, torch.zeros_like(t)).cpu()
    rtp, rfp, rfn = O.tp_fp_fn_hard(x.detach().cpu(), tt)
    # reference formula with the mask applied to every term
    axes = [0] + list(range(2, x.ndim))
    seg = x.detach().cpu().argmax(1)[:, None]
    C = x.shape[1]
    oh = torch.zeros(x.shape).scatter_(1, seg, 1)
    yo = torch.zeros(x.shape, dtype=torch.bool).scatter_(1, tt.long(), 1)
    m = keep.cpu().float()
    assert torch.equal(tp.cpu().float(), (oh * yo * m).sum(axes))
    assert torch.equal(fp.cpu().float(), (oh * (~yo) * m).sum(axes))
    assert torch.equal(fn.cpu().float(), ((1 - oh) * yo * m).sum(axes))


@pytest.mark.parametrize("tag", ["2d", "3d"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_region_loss_matches_reference_golden(hip_lib, tag, dtype):
    """fused sigmoid-Dice + BCE kernel (region-based training) against the reference's DC_and_BCE_loss fixtures, plain and
    with the ignore-mask channel; validation statistics (sigmoid > 0.5) against the oracle formula, exact"""
    import os
    from nnuzoo_amd import hip_ops as ops
    from nnuzoo_amd.training.loss import DC_and_BCE_loss
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"loss_regions_{tag}.npz"))
    bd = bool(g["batch_dice"])
    logits = torch.from_numpy(g["logits"]).to(dtype)
    r, ig = torch.from_numpy(g["regions"]), torch.from_numpy(g["ignore"])
    for name, use, t in (("plain", False, r), ("masked", True, torch.cat([r, ig], 1))):
        loss = DC_and_BCE_loss({}, {'batch_dice': bd, 'do_bg': True, 'smooth': 1e-5, 'ddp': False}, use_ignore_label=use,
                               dice_class=MemoryEfficientSoftDiceLoss)
        x = logits.cuda().requires_grad_(True)
        l = loss(x, t.cuda())
        l.backward()
        if dtype == torch.float32:
            ref_l, ref_g = float(g[f"{name}_loss"]), torch.from_numpy(g[f"{name}_dlogits"])
            ltol, gtol = 2e-5, 1e-4
        else:   # fp16 logits: compare with the oracle on the rounded logits
            xr = logits.float().requires_grad_(True)
            lr_ = O.dc_and_bce(xr, t, bd, use)
            (ref_g,) = torch.autograd.grad(lr_, xr)
            ref_l, ltol, gtol = float(lr_.detach()), 2e-5, 2e-3
        assert abs(float(l) - ref_l) <= ltol * max(1.0, abs(ref_l)), (float(l), ref_l)
        got = x.grad.float().cpu()
        assert torch.allclose(got, ref_g, rtol=gtol, atol=gtol * ref_g.abs().max().item())
        tp, fp, fn = ops.region_tp_fp_fn(logits.cuda(), t.cuda())
        rtp, rfp, rfn = O.region_tp_fp_fn(logits, t, use)
        assert torch.equal(tp.cpu().float(), rtp) and torch.equal(fp.cpu().float(), rfp) and torch.equal(fn.cpu().float(), rfn)


def test_region_trainer_step(hip_lib):
    """nnUNetTrainer with a region-based dataset.json (+ ignore label): heads = foreground regions, DC_and_BCE_loss,
    train and validation step run on one-hot region targets"""
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer
    plans, cfg, dj = nnunet_plans(3, (32, 32, 32), batch_size=2)
    dj = dict(dj)
    dj["labels"] = {"background": 0, "whole": [1, 2, 3], "core": [2, 3], "enh": 3, "ignore": 4}
    dj["regions_class_order"] = [1, 2, 3]
    tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    assert tr.network.num_classes == 3
    scales = tr._get_deep_supervision_scales()
    base = synthetic_batch(2, (32, 32, 32), scales, seed=4)
    g = torch.Generator().manual_seed(1)
    tgt = []
    for t in base["target"]:
        lab = torch.randint(0, 5, t.shape, generator=g)                    # labels 0..3 and the ignore label 4
        reg = torch.cat([((lab >= 1) & (lab <= 3)), ((lab >= 2) & (lab <= 3)), (lab == 3), (lab == 4)], 1).to(torch.int16)
        tgt.append(reg)
    batch = {"data": base["data"], "target": tgt}
    l0 = float(tr.train_step(batch)["loss"])
    l1 = float(tr.train_step(batch)["loss"])
    assert np.isfinite(l0) and np.isfinite(l1)
    v = tr.validation_step(batch)
    assert v["tp_hard"].shape == (3,) and np.isfinite(v["loss"])
    keep = int((tgt[0][:, 3] == 0).sum())
    assert int(v["tp_hard"][0] + v["fn_hard"][0]) == int(((tgt[0][:, 0] == 1) & (tgt[0][:, 3] == 0)).sum()) <= keep


@pytest.mark.parametrize("batch_dice,do_bg,ignore", [(False, False, None), (True, False, None), (False, True, None),
                                                      (True, True, 3), (False, False, 3)])
def test_device_finalize_equals_elementwise_path(hip_lib, monkeypatch, batch_dice, do_bg, ignore):
    """the one-launch loss/coefficients kernel (used whenever batch Dice is not summed across DDP ranks) against the
    element-wise torch arithmetic + autograd it replaces, through the deep-supervision sum and a loss scale"""
    g = torch.Generator().manual_seed(11)
    C, shapes = 3, [(2, 12, 20, 16), (2, 6, 10, 8), (2, 3, 5, 4)]
    weights = [4 / 7, 2 / 7, 1 / 7]
    outs = [(torch.randn(s[0], C, *s[1:], generator=g) * 2).half().cuda() for s in shapes]
    tgs = [torch.randint(0, C + (1 if ignore is not None else 0), (s[0], 1, *s[1:]), generator=g).to(torch.int16).cuda()
           for s in shapes]
    loss = DC_and_CE_loss({'batch_dice': batch_dice, 'smooth': 1e-5, 'do_bg': do_bg, 'ddp': False}, {}, weight_ce=1,
                          weight_dice=1.5, ignore_label=ignore, dice_class=MemoryEfficientSoftDiceLoss)
    w = DeepSupervisionWrapper(loss, weights)
    scale = torch.tensor(1024.0, device="cuda")
    res = []
    for on_device in (True, False):
        monkeypatch.setattr(DC_and_CE_loss, "finalize_on_device", lambda self, v=on_device: v)
        xs = [o.clone().requires_grad_(True) for o in outs]
        l = w(xs, tgs)
        (l * scale).backward()
        res.append((float(l.detach()), [x.grad.float().cpu() for x in xs]))
    assert abs(res[0][0] - res[1][0]) <= 2e-6 * max(1.0, abs(res[1][0])), (res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-3 * b.abs().max().item()), (a - b).abs().max().item()
    # a zero weight drops that output (no gradient), as the reference's `if weights[i] != 0.0`
    monkeypatch.setattr(DC_and_CE_loss, "finalize_on_device", lambda self: True)
    xs = [o.clone().requires_grad_(True) for o in outs]
    DeepSupervisionWrapper(loss, [1.0, 0.0, 0.5])(xs, tgs).backward()
    assert xs[1].grad is None and xs[0].grad is not None and xs[2].grad is not None
