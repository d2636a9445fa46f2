"""Which stage of LM2NetP first disagrees with the reference fixture (tests/golden/net_LM2NetP_64.npz mid_* taps)?
Usage (GPU box): python tools/probes/lm2net_localise.py [LM2NetP|LM2Net]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import det_fill
from nnuzoo_amd.nets import lm2net

name = sys.argv[1] if len(sys.argv) > 1 else "LM2NetP"
z = np.load(os.path.join(ROOT, "tests", "golden", f"net_{name}_64.npz"))
torch.manual_seed(0)
net = getattr(lm2net, name)(spatial_dims=2, in_ch=1, out_ch=2, deep_supervision=True, input_patch_size=(64, 64))
det_fill(net)
with torch.no_grad():
    for n, p in net.named_parameters():
        if n.endswith("A_log"):
            p.copy_(torch.log(1.0 + torch.arange(p.numel(), dtype=torch.float32).reshape(p.shape) % 16) * 0.9 + 0.05 * p)
for m in net.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.zero_()
        m.running_var.fill_(1.0)
net = net.cuda().eval()
mids = {}


def tap(tag):
    def fn(mod, inp, out):
        o = out.detach().reshape(-1).float().cpu()
        mids[f"mid_{tag}"] = np.concatenate([o[::max(1, o.numel() // 64)][:64].numpy(), [float(o.double().pow(2).mean().sqrt())]])
    return fn


for tag, mod in list(net.named_children()) + [(f"stage1.{n}", m) for n, m in net.stage1.named_children()] + \
        [(f"stage1.down_layers.0.1.{n}", m) for n, m in net.stage1.down_layers[0][1].named_children()]:
    mod.register_forward_hook(tap(tag))
outs = net(torch.tensor(z["x"]).cuda())
for k in [k for k in z.files if k.startswith("mid_")]:
    if k not in mids:
        print(f"{k:45s} (not tapped)")
        continue
    r, g = z[k], mids[k]
    print(f"{k:45s} rms ref {r[-1]:10.4g} got {g[-1]:10.4g}   max|d| / rms {np.abs(r[:-1] - g[:-1]).max() / (r[-1] + 1e-12):9.2e}")
