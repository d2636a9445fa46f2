"""BUILD CONTAINER ONLY (needs /root/reference): how well conditioned are the gradients of the reference's LightMUNet under the
deterministic parameter fill of the fixtures?  Runs the reference class twice, the second time with a 1e-6 perturbation of the
input: forward changes 1.5e-4 of the rms, dx by more than its own maximum, parameter-gradient norms by 18 % (median) - i.e. the
fixture's gradient VALUES cannot be reproduced by any other fp32 implementation; tests/test_lightmunet.py therefore compares
the forward tightly and the gradients only in structure and overall scale.
    python tools/probes/lightmunet_reference_conditioning.py"""
import sys, os
import numpy as np, torch
sys.path.insert(0,'/root/repo/tools'); sys.path.insert(0,'/root/repo/tests')
import make_golden_lm2net as M
from golden_util import det_fill
M.bind()
import nnunetv2.nets.LightMUNet as R
def run(eps):
    torch.manual_seed(0)
    net = R.LightMUNet(spatial_dims=2, init_filters=32, in_channels=1, out_channels=3, blocks_down=[1,2,2,4], blocks_up=[1,1,1])
    det_fill(net)
    with torch.no_grad():
        for n,p in net.named_parameters():
            if n.endswith("A_log"):
                p.copy_(torch.log(1.0 + torch.arange(p.numel(), dtype=torch.float32).reshape(p.shape) % 16) * 0.9 + 0.05 * p)
    net.train()
    x = torch.randn(1,1,64,64, generator=torch.Generator().manual_seed(7))
    x = (x + eps*torch.randn(1,1,64,64, generator=torch.Generator().manual_seed(8))).requires_grad_(True)
    y = net(x)
    j = torch.arange(y.numel(), dtype=torch.float64)
    ((y * torch.sin(0.37*j).float().view_as(y)).sum()/y[0,0].numel()).backward()
    g = {n: p.grad.clone() for n,p in net.named_parameters() if p.grad is not None}
    return y.detach(), x.grad.clone(), g
y0,dx0,g0 = run(0.0)
y1,dx1,g1 = run(1e-6)
print("forward change / rms", ((y1-y0).abs().max()/y0.pow(2).mean().sqrt()).item())
print("dx change: max", (dx1-dx0).abs().max().item(), "max ref", dx0.abs().max().item())
rows=sorted(((abs(g1[n].double().norm()-g0[n].double().norm())/(g0[n].double().norm()+1e-30)).item(), n) for n in g0)
print("median rel change of grad norms", rows[len(rows)//2][0]); print(rows[-5:])
