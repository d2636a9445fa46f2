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


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="M2NetP")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--heldout", type=int, default=16)
    a = ap.parse_args()
    print(json.dumps(run(a.model, a.size, a.steps, a.heldout)))
