import sys, torch
sys.path.insert(0, "/root/repo")
from nnuzoo_amd.nets.mamba_simple import MambaSSM
torch.manual_seed(0)
for d_model, L in [(96, 16), (192, 4), (384, 4), (384, 16), (384, 1), (96, 1024), (384, 64)]:
    m = MambaSSM(d_model).cuda()
    x = torch.randn(2, L, d_model, device="cuda", requires_grad=True)
    y = m(x)
    torch.cuda.synchronize()
    print("fwd ok", d_model, L, float(y.abs().max()), flush=True)
    y.pow(2).mean().backward()
    torch.cuda.synchronize()
    print("bwd ok", d_model, L, float(x.grad.abs().max()), flush=True)
