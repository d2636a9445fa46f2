"""Dice protocol, ORACLE side (CPU, travels - no /root/reference): the protocol of tools/dice_ref_cpu_zoo.py with the CPU oracle
nets (oracle/m2net.py, oracle/swt2net.py) in place of the reference's classes - seeded construction through the product's
constructors (bit-identical to the reference's, tests/golden/seeded_init.json; the state_dict loads into the oracle), the same
synthetic batches, AdamW 1e-4 / wd 5e-2 / eps 1e-5, clip 12, DropPath off, foreground Dice on the same held-out patches.
Compared with the reference's own run (--ref-json, e.g. tests/golden/dice_ref_swt2net_128.json) this measures how far two CPU
fp32 runs of the SAME algorithm drift apart over the protocol - the yardstick for the HIP path's Dice gates.
Usage: python tools/dice_oracle_cpu_zoo.py --model SwT2Net --size 128 --steps 200 --ref-json tests/golden/dice_ref_swt2net_128.json \\
           --out profiles/r04_dice_oracle_vs_reference_swt2net_128.json"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.losses import deep_supervision_loss  # noqa: E402
from nnuzoo_amd.synthetic import synthetic_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="SwT2Net", choices=["M2NetP", "SwT2Net"])
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--heldout", type=int, default=16)
ap.add_argument("--ref-json", default="")
ap.add_argument("--out", default="")
ap.add_argument("--threads", type=int, default=0, help="torch CPU threads (0: default); a second run with another count is a "
                "second, equally valid fp32 summation order - the CPU-vs-CPU drift of the protocol")
ap.add_argument("--fixture", default="", help="also write a test fixture (all losses + the packed held-out masks), e.g. "
                "tests/golden/dice_oracle_m2netp_64.json")
a = ap.parse_args()
if a.threads:
    torch.set_num_threads(a.threads)

torch.manual_seed(0)
if a.model == "SwT2Net":
    from oracle.swt2net import SwT2Net as Oracle
    from nnuzoo_amd.nets.swt2net import SwT2Net as Product
else:
    from oracle.m2net import M2NetP as Oracle
    from nnuzoo_amd.nets.m2net import M2NetP as Product
seeded = Product(1, 2, True)
net = Oracle(1, 2, True)
net.load_state_dict(seeded.state_dict())
del seeded
for m in net.modules():
    if hasattr(m, "p") and type(m).__name__ in ("DropPath", "StochasticDepth"):
        m.p = 0.0
scales = [[1.0, 1.0], [1.0, 1.0], [0.5, 0.5], [0.25, 0.25], [0.125, 0.125], [0.0625, 0.0625], [0.03125, 0.03125]]
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=5e-2, eps=1e-5, betas=(0.9, 0.999))
net.train()
losses, t0 = [], time.time()
for it in range(a.steps):
    b = synthetic_batch(2, (a.size, a.size), scales, seed=1000 + it)
    opt.zero_grad(set_to_none=True)
    loss = deep_supervision_loss(list(net(b["data"])), b["target"], batch_dice=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
    opt.step()
    losses.append(float(loss.detach()))
    if it % 10 == 0:
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
        dice += [dice_of(pm[k], gt[k]) for k in range(2)]
res = {"model": a.model + " (CPU oracle, fp32)", "size": a.size, "steps": a.steps, "heldout": a.heldout,
       "dice": float(np.mean(dice)), "losses_first_last": losses[:3] + losses[-3:], "seconds": time.time() - t0}
if a.ref_json:
    ref = json.load(open(a.ref_json))
    assert ref["size"] == a.size and ref["steps"] == a.steps and ref["heldout"] == a.heldout
    import base64
    mine = np.packbits(torch.cat(masks).numpy().reshape(-1))
    theirs = (np.frombuffer(base64.b64decode(ref["masks_packed_b64"]), dtype=np.uint8) if "masks_packed_b64" in ref
              else np.array(ref["masks_packed"], dtype=np.uint8))
    assert theirs.shape == mine.shape, (theirs.shape, mine.shape)
    agree = 1.0 - np.unpackbits(mine ^ theirs).sum() / (8.0 * len(mine))
    n = min(len(losses), len(ref["losses"]))
    res.update({"reference_dice": ref["dice"], "abs_delta": abs(res["dice"] - ref["dice"]), "mask_agreement": float(agree),
                "loss_abs_delta_step0": abs(losses[0] - ref["losses"][0]),
                "loss_abs_delta_max": float(np.max(np.abs(np.array(losses[:n]) - np.array(ref["losses"][:n])))),
                "reference": a.ref_json})
if a.fixture:
    import base64
    fx = {"model": a.model + " (CPU oracle oracle/" + ("swt2net" if a.model == "SwT2Net" else "m2net") + ".py, fp32)", "size": a.size,
          "steps": a.steps, "heldout": a.heldout, "dice": res["dice"], "losses": losses, "threads": torch.get_num_threads(),
          "masks_packed_b64": base64.b64encode(np.packbits(torch.cat(masks).numpy().reshape(-1)).tobytes()).decode(),
          "generator": f"tools/dice_oracle_cpu_zoo.py --model {a.model} --size {a.size} --steps {a.steps} --fixture ..."}
    json.dump(fx, open(a.fixture, "w"))
print(json.dumps(res))
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
