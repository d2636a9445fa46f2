"""Diagnostic (GPU box): eager steps first (optimizer state allocated), then hipGraph replays through the trainer."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training import zoo_trainers as Z

name, n_eager = sys.argv[1], int(sys.argv[2])
plans, cfg, dj = nnunet_plans(2, (512, 512), batch_size=2)
torch.manual_seed(0)
tr = getattr(Z, "nnUNetTrainer" + name)(plans, cfg, 0, dj, device=torch.device("cuda"))
tr.initialize()
b = synthetic_batch(2, (512, 512), tr._get_deep_supervision_scales(), seed=3)
b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
out = []
for it in range(n_eager + 8):
    tr.use_hip_graph = it >= n_eager
    out.append(round(float(tr.train_step(b)["loss"]), 4))
print("RESULT", name, n_eager, out, "scale", tr.grad_scaler.get_scale(), flush=True)
