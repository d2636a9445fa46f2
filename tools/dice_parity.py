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


def run_vs_oracle(fixture):
    """The protocol with the oracle's side read from a committed fixture (tools/dice_oracle_cpu.py, run in the build container:
    the CPU oracle's losses, Dice and packed held-out masks - data) so that it costs seconds and no oracle code runs here: the HIP
    PlainConvUNet (fp16 operands / fp32 accumulate + GradScaler, hipGraph step: the product's train_step) starts from the same
    seeded parameters - constructed on the CPU generator in both places, pinned by the fixture's `init_l2` / `init_abs_first` -
    and sees the same batches."""
    import base64
    ref = json.load(open(fixture))
    edge, steps, heldout = ref["edge"], ref["steps"], ref["heldout"]
    patch = (edge,) * 3
    plans, cfg, dj = nnunet_plans(3, patch, batch_size=2)
    torch.manual_seed(0)
    tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    state = tr.network.state_dict()
    l2 = float(sum(v.double().pow(2).sum() for v in state.values()).sqrt())
    first = float(next(iter(state.values())).double().abs().sum())
    assert abs(l2 - ref["init_l2"]) <= 1e-6 * ref["init_l2"] and abs(first - ref["init_abs_first"]) <= 1e-6 * ref["init_abs_first"], \
        ("seeded construction differs from the fixture's", l2, ref["init_l2"], first, ref["init_abs_first"])
    scales = tr._get_deep_supervision_scales()
    lh = []
    for it in range(steps):
        lh.append(float(tr.train_step(synthetic_batch(2, patch, scales, seed=1000 + it))["loss"]))
    tr.network.eval()
    dh, masks = [], []
    with torch.no_grad():
        for i in range(heldout // 2):
            b = synthetic_batch(2, patch, scales, seed=90000 + i)
            gt = b["target"][0][:, 0]
            ph = tr.network(b["data"].cuda())[0].float().cpu().argmax(1)
            masks.append(ph.to(torch.uint8))
            dh += [dice_of(ph[k], gt[k]) for k in range(2)]
    mine = np.packbits(torch.cat(masks).numpy().reshape(-1))
    theirs = np.frombuffer(base64.b64decode(ref["masks_packed_b64"]), dtype=np.uint8)
    agree = 1.0 - np.unpackbits(mine ^ theirs).sum() / (8.0 * len(mine))
    n = min(len(lh), len(ref["losses"]))
    return {"edge": edge, "steps": steps, "heldout": heldout, "dice_hip": float(np.mean(dh)), "dice_oracle": ref["dice"],
            "abs_delta": abs(float(np.mean(dh)) - ref["dice"]), "mask_agreement": float(agree),
            "loss_abs_delta_step0": abs(lh[0] - ref["losses"][0]),
            "loss_abs_delta_max": float(np.max(np.abs(np.array(lh[:n]) - np.array(ref["losses"][:n])))),
            "loss_hip_last": lh[-1], "loss_oracle_last": ref["losses"][-1], "oracle": os.path.basename(fixture)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle-json", default="", help="tests/golden/dice_oracle_plainconv_64.json: HIP side only, oracle from the fixture")
    ap.add_argument("--edge", type=int, default=32)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--heldout", type=int, default=16)
    a = ap.parse_args()
    print(json.dumps(run_vs_oracle(a.oracle_json) if a.oracle_json else run(a.edge, a.steps, a.heldout)))
