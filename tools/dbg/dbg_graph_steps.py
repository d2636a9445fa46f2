"""Diagnostic (GPU box): per-step loss scale / gradient norm / finiteness of the hipGraph-replayed zoo step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training import zoo_trainers as Z
from nnuzoo_amd.training.graph_step import GraphedForwardBackward

name = sys.argv[1] if len(sys.argv) > 1 else "M2NetP"
graph = int(sys.argv[2]) if len(sys.argv) > 2 else 1
plans, cfg, dj = nnunet_plans(2, (512, 512), batch_size=2)
torch.manual_seed(0)
tr = getattr(Z, "nnUNetTrainer" + name)(plans, cfg, 0, dj, device=torch.device("cuda"))
tr.initialize()
b = synthetic_batch(2, (512, 512), tr._get_deep_supervision_scales(), seed=3)
data, target = b["data"].cuda(), [t.cuda() for t in b["target"]]
g = GraphedForwardBackward(tr.network, tr.loss, tr.grad_scaler, autocast=True)
params = [p for p in tr.network.parameters()]
for it in range(9):
    if graph:
        l = g(data, target)
    else:
        tr.optimizer.zero_grad(set_to_none=True)
        l = g._fwd_bwd(data, target)
    torch.cuda.synchronize()
    sc = tr.grad_scaler.get_scale()
    gs = [p.grad for p in params if p.grad is not None]
    nonfin = sum(int((~torch.isfinite(x)).any()) for x in gs)
    if it >= 5:
        print("  nonfinite:", [n for n, p in tr.network.named_parameters()
                               if p.grad is not None and not bool(torch.isfinite(p.grad).all())][:12], flush=True)
    gn = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(x.float()) for x in gs])).item() / sc
    tr.grad_scaler.unscale_(tr.optimizer)
    torch.nn.utils.clip_grad_norm_(tr.network.parameters(), 12)
    tr.grad_scaler.step(tr.optimizer)
    tr.grad_scaler.update()
    pbad = sum(int((~torch.isfinite(p)).any()) for p in params)
    print(f"it {it} loss {l.item():.4f} scale {sc} grads {len(gs)} nonfinite-grad-tensors {nonfin} gnorm {gn:.4g} "
          f"nonfinite-params {pbad}", flush=True)
