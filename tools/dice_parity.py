"""Dice parity protocol of SURVEY.md §8d at a reduced shape the CPU oracle can finish: train the HIP PlainConvUNet
(GPU, fp16 operands / fp32 accumulate + GradScaler) and the CPU oracle (fp32) from the SAME seeded weights on the
SAME synthetic batches with the reference's optimiser settings, then evaluate foreground Dice = 2TP/(2TP+FP+FN)
(nnUNetTrainer.py:1255) on held-out synthetic patches.  Prints one JSON line.
Usage: python tools/dice_parity.py [--edge 32] [--steps 60] [--heldout 16]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.losses import deep_supervision_loss
from oracle.plain_conv_unet import OraclePlainConvUNet, planner_arch_kwargs
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer


def dice_of(pred_mask, gt):
    tp = ((pred_mask == 1) & (gt == 1)).sum().item()
    fp = ((pred_mask == 1) & (gt == 0)).sum().item()
    fn = ((pred_mask == 0) & (gt == 1)).sum().item()
    return 2 * tp / max(1, 2 * tp + fp + fn)


def run(edge=32, steps=60, heldout=16, seed=0):
    patch = (edge,) * 3
    plans, cfg, dj = nnunet_plans(3, patch, batch_size=2)
    arch = plans["configurations"][cfg]["architecture"]["arch_kwargs"]
    torch.manual_seed(seed)
    tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    ref = OraclePlainConvUNet(1, num_classes=2, **planner_arch_kwargs(3, arch["n_stages"], arch["features_per_stage"]))
    ref.load_state_dict({k: v.cpu() for k, v in tr.network.state_dict().items()})
    opt = torch.optim.SGD(ref.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    scales = tr._get_deep_supervision_scales()
    lh, lr = [], []
    for it in range(steps):
        b = synthetic_batch(2, patch, scales, seed=1000 + it)
        lh.append(float(tr.train_step(b)["loss"]))
        opt.zero_grad()
        l = deep_supervision_loss(ref(b["data"]), b["target"], batch_dice=False)
        l.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 12)
        opt.step()
        lr.append(l.item())
    tr.network.eval()
    ref.eval()
    dh, dr, agree = [], [], []
    with torch.no_grad():
        for i in range(heldout // 2):
            b = synthetic_batch(2, patch, scales, seed=90000 + i)
            gt = b["target"][0][:, 0]
            ph = tr.network(b["data"].cuda())[0].float().cpu().argmax(1)
            pr = ref(b["data"])[0].argmax(1)
            for k in range(2):
                dh.append(dice_of(ph[k], gt[k]))
                dr.append(dice_of(pr[k], gt[k]))
            agree.append((ph == pr).float().mean().item())
    return {"edge": edge, "steps": steps, "heldout": heldout, "dice_hip": float(np.mean(dh)), "dice_oracle": float(np.mean(dr)),
            "abs_delta": abs(float(np.mean(dh)) - float(np.mean(dr))), "mask_agreement": float(np.mean(agree)),
            "loss_hip_last": lh[-1], "loss_oracle_last": lr[-1]}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--edge", type=int, default=32)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--heldout", type=int, default=16)
    a = ap.parse_args()
    print(json.dumps(run(a.edge, a.steps, a.heldout)))
