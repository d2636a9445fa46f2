"""Diagnostic (GPU box): hipGraph replays of the zoo step with / without the eager optimizer part in between."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training import zoo_trainers as Z
from nnuzoo_amd.training.graph_step import GraphedForwardBackward

name, mode = sys.argv[1], sys.argv[2]
plans, cfg, dj = nnunet_plans(2, (512, 512), batch_size=2)
torch.manual_seed(0)
tr = getattr(Z, "nnUNetTrainer" + name)(plans, cfg, 0, dj, device=torch.device("cuda"))
tr.initialize()
b = synthetic_batch(2, (512, 512), tr._get_deep_supervision_scales(), seed=3)
data, target = b["data"].cuda(), [t.cuda() for t in b["target"]]
g = GraphedForwardBackward(tr.network, tr.loss, tr.grad_scaler, autocast=True)
out = []
for it in range(6):
    l = g(data, target)
    if mode == "opt":
        tr.grad_scaler.unscale_(tr.optimizer)
        torch.nn.utils.clip_grad_norm_(tr.network.parameters(), 12)
        tr.grad_scaler.step(tr.optimizer)
        tr.grad_scaler.update()
    elif mode == "trample":     # eager allocations + writes between replays, no optimizer
        junk = [torch.full((16 * 1024 * 1024,), float("nan"), device="cuda") for _ in range(8 + it)]
        torch.cuda.synchronize()
        del junk
    elif mode == "perturb":     # change every parameter in place by hand (no optimizer, no temporaries)
        with torch.no_grad():
            torch._foreach_add_([p for p in tr.network.parameters()], 1e-5)
    elif mode == "smalljunk":   # many small eager allocations written with NaN (what foreach optimizers create)
        junk = [torch.full((1 + 37 * i,), float("nan"), device="cuda") for i in range(3000)]
        torch.cuda.synchronize()
        del junk
    elif mode == "scale":       # only lower the loss scale by hand, no optimizer
        tr.grad_scaler._scale.mul_(0.5)
    out.append(round(float(l.detach().cpu()), 4))
print("RESULT", name, mode, out, "scale", tr.grad_scaler.get_scale(), flush=True)
