"""Which module of an M2NetP forward differs between two passes on the same input (training mode, same DropPath seed)?"""
import torch
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerM2NetP

plans, cfg, dj = nnunet_plans(2, (128, 128), batch_size=2)
torch.manual_seed(0)
tr = nnUNetTrainerM2NetP(plans, cfg, 0, dj, device=torch.device("cuda"))
tr.initialize()
net = tr.network
b = synthetic_batch(2, (128, 128), tr._get_deep_supervision_scales(), seed=100)
x = b["data"].cuda()
rec = [[], []]
cur = [0]


def hook(name):
    def fn(mod, a, out):
        if torch.is_tensor(out):
            rec[cur[0]].append((name, out.detach().float().clone()))
    return fn


for n, m in net.named_modules():
    if n:
        m.register_forward_hook(hook(n))
for k in range(2):
    cur[0] = k
    torch.manual_seed(1)
    with torch.autocast("cuda", dtype=torch.float16):
        outs = net(x)
    torch.cuda.synchronize()
print(len(rec[0]), len(rec[1]))
bad = 0
for (n0, a), (n1, c) in zip(*rec):
    assert n0 == n1
    if not torch.equal(a, c):
        print("DIFF", n0, float((a - c).abs().max()), float(a.abs().max()))
        bad += 1
        if bad > 12:
            break
print("bad", bad)
