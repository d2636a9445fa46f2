import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from nnuzoo_amd import conv_plan as cp, hip_ops as ops
from nnuzoo_amd.hip_ops import PreparedTable
DEV='cuda'
g=torch.Generator().manual_seed(0)
def cl(x): N,C=x.shape[:2]; return x.permute(0,2,3,4,1).reshape(N,-1,C).contiguous().half().to(DEV)
for (N,dims,cin,cout,stride) in [(2,(8,8,8),64,128,2),(2,(4,4,4),128,128,1),(2,(8,8,8),128,256,2),(2,(4,4,4),256,256,1),(2,(16,16,16),32,64,2),(2,(32,32,32),32,32,1)]:
    x=torch.randn(N,cin,*dims,generator=g); w=(torch.randn(cout,cin,3,3,3,generator=g)*0.05).to(DEV)
    pt=PreparedTable(cp.conv_forward(N,dims,cin,cout,stride=stride))
    wp=ops.pack_weight(w,pt,cin,cout,27,cin*27,1)
    od=pt.table.out_dims; V=int(np.prod(od))
    outs=[];sts=[]
    for rep in range(3):
        out=torch.empty((N,V,cout),dtype=torch.float16,device=DEV); st=torch.zeros((N,cout,2),device=DEV)
        ops.conv_tap_forward(pt,cl(x),wp,None,out,stats=st); torch.cuda.synchronize()
        outs.append(out);sts.append(st)
    st_ref=torch.stack([outs[0].double().sum(1),(outs[0].double()**2).sum(1)],-1)
    print("fwd",dims,cin,cout,stride,"out equal",torch.equal(outs[0],outs[1]),torch.equal(outs[0],outs[2]),"stats relerr",((sts[0].double()-st_ref).abs().max()/st_ref.abs().max()).item(), ((sts[1]-sts[0]).abs().max()/sts[0].abs().max()).item())
    # dgrad
    dy=torch.randn(N,cout,*od,generator=g)
    ptd=PreparedTable(cp.conv_dgrad(N,dims,cin,cout,stride=stride)); wpd=ops.pack_weight(w,ptd,cout,cin,cin*27,27,1)
    dxs=[]
    for rep in range(3):
        dx=torch.empty((N,int(np.prod(dims)),cin),dtype=torch.float16,device=DEV)
        ops.conv_tap_forward(ptd,cl(dy),wpd,None,dx); torch.cuda.synchronize(); dxs.append(dx)
    print("   dgrad equal",torch.equal(dxs[0],dxs[1]),torch.equal(dxs[0],dxs[2]))
    ptw=PreparedTable(cp.conv_wgrad(N,dims,cin,cout,stride=stride))
    dws=[]
    for rep in range(3):
        dw=torch.empty((27,cin,cout),device=DEV); ops.conv_tap_wgrad(ptw,cl(x),cl(dy),dw); torch.cuda.synchronize(); dws.append(dw)
    print("   wgrad rel diff",((dws[0]-dws[1]).norm()/dws[0].norm()).item(),((dws[0]-dws[2]).norm()/dws[0].norm()).item())
