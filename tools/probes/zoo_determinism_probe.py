"""probe: which parameter gradients of a zoo network differ between two identical forward + backward passes?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nnuzoo_amd.nets import m2net, swt2net

for name, cls, autocast in [("SwT2Net", swt2net.SwT2Net, False), ("M2NetP", m2net.M2NetP, True)]:
    torch.manual_seed(0)
    net = cls(1, 2, True).cuda().train()
    x = torch.randn(2, 1, 128, 128, device="cuda")
    grads = []
    for rep in range(2):
        net.zero_grad(set_to_none=True)
        torch.manual_seed(5)
        with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
            outs = net(x)
        sum((o.float() ** 2).mean() for o in outs).backward()
        grads.append({n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
    diff = [n for n in grads[0] if not torch.equal(grads[0][n], grads[1][n])]
    fam = {}
    for n in diff:
        k = ".".join(n.split(".")[-3:])
        fam[k] = fam.get(k, 0) + 1
    print(name, "params with grad", len(grads[0]), "differ", len(diff))
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:25]:
        print("   ", v, k)
