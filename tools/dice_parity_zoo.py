"""Dice protocol for the SS2D^2Net path: train M2NetP on the GPU twice from the SAME seeded weights on the SAME synthetic
batches - (a) the product path (fused SS2D block: dwconv+SiLU+layouts, cross-scan chunk scan, merge, gated LayerNorm; HIP
LayerNorm everywhere) and (b) the reference's op-by-op formulation of the block (stack / flip / einsum / selective_scan_fn /
flip / add, torch layer_norm), which the golden vectors pin to the reference - then compare the foreground Dice
2TP/(2TP+FP+FN) on held-out synthetic patches.  DropPath is disabled in both runs (its RNG stream would differ).
Prints one JSON line.   Usage: python tools/dice_parity_zoo.py [--size 128] [--steps 80] [--heldout 16] [--model M2NetP]"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnuzoo_amd import layer_norm as LN
from nnuzoo_amd.nets.m2net import SS2D
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training import zoo_trainers as Z


def dice_of(pred_mask, gt):
    tp = ((pred_mask == 1) & (gt == 1)).sum().item()
    fp = ((pred_mask == 1) & (gt == 0)).sum().item()
    fn = ((pred_mask == 0) & (gt == 1)).sum().item()
    return 2 * tp / max(1, 2 * tp + fp + fn)


def _train(model, size, steps, heldout, fused, seed):
    hip_ln = LN.LayerNorm.forward
    SS2D.fused_cross_scan = fused
    if not fused:
        LN.LayerNorm.forward = lambda self, x: F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
    try:
        plans, cfg, dj = nnunet_plans(2, (size, size), batch_size=2)
        torch.manual_seed(seed)
        tr = getattr(Z, "nnUNetTrainer" + model)(plans, cfg, 0, dj, device=torch.device("cuda"))
        tr.initialize()
        for m in tr.network.modules():
            if hasattr(m, "drop_prob"):
                m.drop_prob = 0.0
        scales = tr._get_deep_supervision_scales()
        losses = []
        for it in range(steps):
            b = synthetic_batch(2, (size, size), scales, seed=1000 + it)
            losses.append(float(tr.train_step(b)["loss"]))
        tr.network.eval()
        dice, masks = [], []
        with torch.no_grad(), torch.autocast("cuda"):
            for i in range(heldout // 2):
                b = synthetic_batch(2, (size, size), scales, seed=90000 + i)
                gt = b["target"][0][:, 0]
                out = tr.network(b["data"].cuda())
                pm = (out[0] if isinstance(out, (tuple, list)) else out).float().cpu().argmax(1)
                masks.append(pm)
                for k in range(2):
                    dice.append(dice_of(pm[k], gt[k]))
        return float(np.mean(dice)), torch.cat(masks), losses
    finally:
        SS2D.fused_cross_scan = True
        LN.LayerNorm.forward = hip_ln


def run(model="M2NetP", size=128, steps=80, heldout=16, seed=0):
    d_ref, m_ref, l_ref = _train(model, size, steps, heldout, fused=False, seed=seed)
    d_hip, m_hip, l_hip = _train(model, size, steps, heldout, fused=True, seed=seed)
    return {"model": model, "size": size, "steps": steps, "heldout": heldout, "dice_fused": d_hip,
            "dice_reference_formulation": d_ref, "abs_delta": abs(d_hip - d_ref),
            "mask_agreement": (m_hip == m_ref).float().mean().item(), "loss_fused_last": l_hip[-1],
            "loss_reference_last": l_ref[-1], "loss_first": l_ref[0]}


def run_vs_oracle(fixture, autocast=False):
    """SURVEY.md 8d Dice protocol with the CPU ORACLE on the other side: `fixture` =
    tests/golden/dice_oracle_m2netp_64.json, written in the build container by tools/dice_oracle_cpu_zoo.py - oracle/m2net.py
    (pinned by the reference's own outputs, gradients and 6-step training trajectory, tests/test_oracle_m2net.py) trained in fp32
    from the seeded construction, its foreground Dice on the held-out patches, every loss and its argmax masks.  Here the HIP
    M2NetP repeats the protocol on the GPU: same seeded parameters, same batches, HIP loss, AdamW 1e-4 / wd 5e-2 / eps 1e-5,
    clip 12, DropPath off, fp32 (autocast=True: the product's fp16-autocast + GradScaler step, whose first updates are skipped
    while the loss scale backs off - reported, not gated)."""
    import base64
    from nnuzoo_amd.nets.m2net import M2NetP
    from nnuzoo_amd.training.loss import DC_and_CE_loss, DeepSupervisionWrapper, MemoryEfficientSoftDiceLoss
    ref = json.load(open(fixture))
    size, steps, heldout = ref["size"], ref["steps"], ref["heldout"]
    torch.manual_seed(0)
    net = M2NetP(1, 2, True)
    for m in net.modules():
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
    net = net.cuda().train()
    scales = [[1.0, 1.0], [1.0, 1.0], [0.5, 0.5], [0.25, 0.25], [0.125, 0.125], [0.0625, 0.0625], [0.03125, 0.03125]]
    w = np.array([1 / (2 ** i) for i in range(len(scales))])
    w[-1] = 0
    w = w / w.sum()
    loss_fn = DeepSupervisionWrapper(DC_and_CE_loss({'batch_dice': True, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {},
                                                    weight_ce=1, weight_dice=1, ignore_label=None,
                                                    dice_class=MemoryEfficientSoftDiceLoss), w)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=5e-2, eps=1e-5, betas=(0.9, 0.999))
    scaler = torch.amp.GradScaler("cuda") if autocast else None
    losses = []
    for it in range(steps):
        b = synthetic_batch(2, (size, size), scales, seed=1000 + it)
        data, target = b["data"].cuda(), [t.cuda() for t in b["target"]]
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", enabled=autocast):
            l = loss_fn(list(net(data)), target)
        if autocast:
            scaler.scale(l).backward()
            scaler.unscale_(opt)
            torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
            scaler.step(opt)
            scaler.update()
        else:
            l.backward()
            torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
            opt.step()
        losses.append(float(l.detach()))
    net.eval()
    dice, masks = [], []
    with torch.no_grad(), torch.autocast("cuda", enabled=autocast):
        for i in range(heldout // 2):
            b = synthetic_batch(2, (size, size), scales, seed=90000 + i)
            gt = b["target"][0][:, 0]
            pm = net(b["data"].cuda())[0].float().cpu().argmax(1)
            masks.append(pm.to(torch.uint8))
            dice += [dice_of(pm[k], gt[k]) for k in range(2)]
    mine = np.packbits(torch.cat(masks).numpy().reshape(-1))
    theirs = np.frombuffer(base64.b64decode(ref["masks_packed_b64"]), dtype=np.uint8)
    agree = 1.0 - np.unpackbits(mine ^ theirs).sum() / (8.0 * len(mine))
    d = float(np.mean(dice))
    n = min(len(losses), len(ref["losses"]))
    return {"model": "M2NetP", "size": size, "steps": steps, "heldout": heldout, "precision": "fp16 autocast" if autocast else "fp32",
            "dice_hip": d, "dice_oracle": ref["dice"], "abs_delta": abs(d - ref["dice"]), "mask_agreement": float(agree),
            "loss_abs_delta_step0": abs(losses[0] - ref["losses"][0]),
            "loss_abs_delta_max": float(np.max(np.abs(np.array(losses[:n]) - np.array(ref["losses"][:n])))),
            "loss_hip_last": losses[-1], "loss_oracle_last": ref["losses"][-1], "oracle": os.path.basename(fixture)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="M2NetP")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--heldout", type=int, default=16)
    ap.add_argument("--oracle-json", default="", help="tests/golden/dice_oracle_m2netp_64.json: the protocol against the CPU oracle")
    ap.add_argument("--autocast", type=int, default=0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    r = run_vs_oracle(a.oracle_json, bool(a.autocast)) if a.oracle_json else run(a.model, a.size, a.steps, a.heldout)
    print(json.dumps(r))
    if a.out:
        json.dump(r, open(a.out, "w"), indent=1)
