import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer
plans, cfg, dj = nnunet_plans(3, (32,32,32), batch_size=2)
torch.manual_seed(0)
tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda")); tr.initialize()
batch = synthetic_batch(2, (32,32,32), tr._get_deep_supervision_scales(), seed=3)
data = batch['data'].cuda(); target=[t.cuda() for t in batch['target']]
def grads():
    tr.optimizer.zero_grad(set_to_none=True)
    out = tr.network(data); l = tr.loss(out, target); (l*65536).backward()
    return {n: p.grad.detach().clone() for n,p in tr.network.named_parameters()}
g1=grads(); g2=grads()
w=[]
for n in g1:
    d=(g1[n]-g2[n]).norm().item(); b=g2[n].norm().item()
    w.append((d/(b+1e-12), n, b))
w.sort(reverse=True)
for x in w[:10]: print(x)
