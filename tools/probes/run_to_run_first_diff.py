import sys, torch, functools
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer
from nnuzoo_amd import hip_ops as ops
import nnuzoo_amd.nets.plain_conv_unet as pcu
plans, cfg, dj = nnunet_plans(3, (32,32,32), batch_size=2)
torch.manual_seed(0)
tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda")); tr.initialize()
batch = synthetic_batch(2, (32,32,32), tr._get_deep_supervision_scales(), seed=3)
data = batch['data'].cuda(); target=[t.cuda() for t in batch['target']]
log=[]
def wrap(name, out_idx):
    orig=getattr(ops,name)
    @functools.wraps(orig)
    def f(*a, **k):
        r=orig(*a, **k)
        torch.cuda.synchronize()
        outs=[]
        for i in out_idx:
            t = a[i] if isinstance(i,int) else k.get(i)
            if t is not None: outs.append(t.detach().float().clone())
        log.append((name, outs))
        return r
    setattr(ops,name,f)
wrap('conv_tap_forward',[4,'stats'])
wrap('conv_tap_wgrad',[3])
wrap('instnorm_lrelu_apply',[4])
wrap('instnorm_lrelu_bwd',[5,6])
wrap('stem_forward',[3]); wrap('stem_wgrad',[2])
wrap('head_forward',[3]); wrap('head_dgrad',[2]); wrap('head_wgrad',[2,3])
wrap('instnorm_stats',[1])
def run():
    log.clear()
    tr.optimizer.zero_grad(set_to_none=True)
    out = tr.network(data); l = tr.loss(out, target); (l*65536).backward()
    torch.cuda.synchronize()
    return list(log)
a=run(); b=run()
print(len(a),len(b))
for i,((n1,o1),(n2,o2)) in enumerate(zip(a,b)):
    for j,(x,y) in enumerate(zip(o1,o2)):
        d=(x-y).norm().item()/(y.norm().item()+1e-12)
        if d>1e-4 or not torch.isfinite(x).all():
            print("op",i,n1,"out",j,"rel diff",d,"shape",tuple(x.shape),"finite",bool(torch.isfinite(x).all()))
