"""probe: which static gradients does a graph replay NOT rewrite after an eager step in between?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerM2NetP
plans, cfg, dj = nnunet_plans(2, (64, 64), batch_size=2)
torch.manual_seed(0)
tr = nnUNetTrainerM2NetP(plans, cfg, 0, dj, device=torch.device("cuda"))
tr.initialize()
b = synthetic_batch(2, (64, 64), tr._get_deep_supervision_scales(), seed=5)
b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
for _ in range(2):
    print("graph loss", float(tr.train_step(b)["loss"]))
names = {id(p): n for n, p in tr.network.named_parameters()}
G = tr._graphed
def probe(tag):
    for p, g in G._static_grads:
        g.fill_(float("nan"))
    torch.cuda.synchronize()
    l = G(b["data"], b["target"])
    torch.cuda.synchronize()
    bad = [names[id(p)] for p, g in G._static_grads if not bool(torch.isfinite(g).all())]
    print(tag, "loss", float(l), "static grads", len(G._static_grads), "non-finite after replay", len(bad), bad[:8])
probe("before eager step:")
tr.use_hip_graph = False
print("eager loss", float(tr.train_step(b)["loss"]))
tr.use_hip_graph = True
probe("after eager step:")
