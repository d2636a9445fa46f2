"""Dice protocol, REFERENCE side (build container only - needs /root/reference): trains the reference's own M2NetP
(nnunetv2/nets/m2net.py imported through tools/ref_shim.py: selective_scan_fn := the reference's selective_scan_ref) on the
CPU in fp32 with the reference's own loss classes, from the seeded construction (bit-identical to the product's, see
tests/golden/seeded_init.json), on the same synthetic batches as tools/dice_parity_zoo.py, with the plugin's optimiser
settings (nnUNetTrainerM2Net.py:58-65: AdamW 1e-4 / wd 5e-2 / eps 1e-5, clip 12; DropPath off as in the HIP run), and
evaluates the foreground Dice on the same held-out patches.  Writes profiles/<tag>.json; the HIP side
(tools/dice_parity_zoo.py --ref-json) compares against it.
Usage: python tools/dice_ref_cpu_zoo.py --size 64 --steps 40 --out profiles/r02_dice_ref_cpu_m2netp_64.json"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_shim  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="M2NetP", choices=["M2NetP", "SwT2Net"])
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--heldout", type=int, default=16)
ap.add_argument("--out", default="")
a = ap.parse_args()
ref_shim.install()
from nnunetv2.nets import m2net, swt2net  # noqa: E402
from nnunetv2.training.loss.compound_losses import DC_and_CE_loss  # noqa: E402
from nnunetv2.training.loss.deep_supervision import DeepSupervisionWrapper  # noqa: E402
from nnunetv2.training.loss.dice import MemoryEfficientSoftDiceLoss  # noqa: E402
from nnuzoo_amd.synthetic import synthetic_batch  # noqa: E402  (the batch generator is shared: same seeds, same voxels)

torch.manual_seed(0)
net = {"M2NetP": m2net.M2NetP, "SwT2Net": swt2net.SwT2Net}[a.model](1, 2, True)
for m in net.modules():
    if hasattr(m, "drop_prob"):
        m.drop_prob = 0.0
scales = [[1.0, 1.0], [1.0, 1.0], [0.5, 0.5], [0.25, 0.25], [0.125, 0.125], [0.0625, 0.0625], [0.03125, 0.03125]]
w = np.array([1 / (2 ** i) for i in range(len(scales))])
w[-1] = 0
w = w / w.sum()                                                       # nnUNetTrainer._build_loss, nnUNetTrainer.py:473-487
loss_fn = DeepSupervisionWrapper(DC_and_CE_loss({'batch_dice': True, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {},
                                                weight_ce=1, weight_dice=1, ignore_label=None,
                                                dice_class=MemoryEfficientSoftDiceLoss), w)
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=5e-2, eps=1e-5, betas=(0.9, 0.999))
net.train()
losses = []
t0 = time.time()
for it in range(a.steps):
    b = synthetic_batch(2, (a.size, a.size), scales, seed=1000 + it)
    opt.zero_grad(set_to_none=True)
    out = net(b["data"])
    l = loss_fn(list(out), b["target"])
    l.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
    opt.step()
    losses.append(float(l))
    print(f"step {it} loss {losses[-1]:.4f} ({time.time() - t0:.0f} s)", flush=True)
net.eval()


def dice_of(pm, gt):
    tp = ((pm == 1) & (gt == 1)).sum().item()
    fp = ((pm == 1) & (gt == 0)).sum().item()
    fn = ((pm == 0) & (gt == 1)).sum().item()
    return 2 * tp / max(1, 2 * tp + fp + fn)


dice, masks = [], []
with torch.no_grad():
    for i in range(a.heldout // 2):
        b = synthetic_batch(2, (a.size, a.size), scales, seed=90000 + i)
        gt = b["target"][0][:, 0]
        pm = net(b["data"])[0].argmax(1)
        masks.append(pm.to(torch.uint8))
        for k in range(2):
            dice.append(dice_of(pm[k], gt[k]))
res = {"model": a.model + " (reference classes, CPU fp32)", "size": a.size, "steps": a.steps,
       "heldout": a.heldout, "dice": float(np.mean(dice)), "losses": losses, "seconds": time.time() - t0,
       "masks_packed": np.packbits(torch.cat(masks).numpy().reshape(-1)).tolist() if a.size <= 128 else None}
if a.out:
    json.dump(res, open(a.out, "w"))
print(json.dumps({k: v for k, v in res.items() if k != "masks_packed"}))
