"""Diagnostic (GPU box): which building block's hipGraph replay reads memory outside the graph's pool.  Each candidate is
captured alone (forward + backward), replayed, then many small eager tensors are filled with NaN and it is replayed
again; outputs / gradients that change point at the culprit."""
import os
import sys

import torch
import torch.nn.functional as F
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nnuzoo_amd.layer_norm import LayerNorm
from nnuzoo_amd.nets import m2net as M
from nnuzoo_amd.nets.common2d import DropPath, PatchExpand, PatchMerging2D, REBNCONV
from nnuzoo_amd.training.loss import DC_and_CE_loss, DeepSupervisionWrapper, MemoryEfficientSoftDiceLoss


def junk():
    j = [torch.full((1 + 37 * i,), float("nan"), device="cuda") for i in range(3000)]
    torch.cuda.synchronize()
    del j


def check(name, mod, make_in, autocast=True, scaler=None, loss_mode=False):
    torch.manual_seed(0)
    ins = make_in()
    params = [p for p in mod.parameters()] if isinstance(mod, nn.Module) else []

    def run():
        with torch.autocast("cuda", enabled=autocast):
            out = mod(*ins)
        o = out if not isinstance(out, (tuple, list)) else out[0]
        l = o.float().sum() if not loss_mode else o
        (scaler.scale(l) if scaler is not None else l).backward()
        return o

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            for p in params:
                p.grad = None
            for t in ins:
                if torch.is_tensor(t) and t.requires_grad:
                    t.grad = None
            run()
    torch.cuda.current_stream().wait_stream(side)
    for p in params:
        p.grad = None
    for t in ins:
        if torch.is_tensor(t) and t.requires_grad:
            t.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = run()
    res = []
    for rep in range(3):
        g.replay()
        torch.cuda.synchronize()
        vals = [out.detach().float().clone()] + [p.grad.detach().float().clone() for p in params if p.grad is not None] + \
               [t.grad.detach().float().clone() for t in ins if torch.is_tensor(t) and t.requires_grad and t.grad is not None]
        res.append(vals)
        junk()
    bad = []
    for rep in (1, 2):
        for i, (a, b) in enumerate(zip(res[0], res[rep])):
            same = torch.equal(torch.nan_to_num(a, nan=123.0), torch.nan_to_num(b, nan=123.0)) or \
                torch.allclose(a, b, rtol=1e-3, atol=1e-5 * max(a.abs().max().item(), 1e-30), equal_nan=True)
            if not same:
                bad.append((rep, i))
    print("CHECK", name, "OK" if not bad else f"CHANGED {bad[:6]}", flush=True)


dev = "cuda"
rin = lambda *s, dtype=torch.float32: torch.randn(*s, device=dev, dtype=dtype).requires_grad_(True)
sc = torch.amp.GradScaler("cuda")
which = sys.argv[1].split(",") if len(sys.argv) > 1 else None
tests = {
    "layernorm": lambda: check("layernorm", LayerNorm(32).cuda(), lambda: [rin(2, 24, 24, 32)]),
    "linear": lambda: check("linear", nn.Linear(32, 64).cuda(), lambda: [rin(2, 24, 24, 32)]),
    "ss2d_fused": lambda: check("ss2d_fused", M.SS2D(d_model=16).cuda(), lambda: [rin(2, 24, 24, 16)]),
    "vssblock": lambda: check("vssblock", M.VSSBlock(hidden_dim=16, drop_path=0.1).cuda(), lambda: [rin(2, 24, 24, 16)]),
    "droppath": lambda: check("droppath", DropPath(0.2), lambda: [rin(8, 24, 24, 16)]),
    "rebnconv": lambda: check("rebnconv", REBNCONV(16, 16).cuda(), lambda: [rin(2, 16, 32, 32)]),
    "merge": lambda: check("merge", PatchMerging2D(16, 2, 32).cuda(), lambda: [rin(2, 24, 24, 16)]),
    "expand": lambda: check("expand", PatchExpand(32, 2).cuda(), lambda: [rin(2, 12, 12, 32)]),
    "interp": lambda: check("interp", lambda x: F.interpolate(x, size=(48, 48), mode="bilinear"), lambda: [rin(2, 4, 24, 24)]),
    "scaler": lambda: check("scaler", nn.Linear(32, 64).cuda(), lambda: [rin(2, 24, 24, 32)], scaler=sc),
    "loss": lambda: check("loss", DeepSupervisionWrapper(DC_and_CE_loss(
        {'batch_dice': True, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {}, weight_ce=1, weight_dice=1,
        ignore_label=None, dice_class=MemoryEfficientSoftDiceLoss), [0.6, 0.4]),
        lambda: [[rin(2, 2, 32, 32, dtype=torch.float16), rin(2, 2, 16, 16, dtype=torch.float16)],
                 [torch.randint(0, 2, (2, 1, 32, 32), device=dev).to(torch.int16),
                  torch.randint(0, 2, (2, 1, 16, 16), device=dev).to(torch.int16)]], loss_mode=True),
}
for k, fn in tests.items():
    if which is None or k in which:
        try:
            fn()
        except Exception as e:  # noqa
            print("CHECK", k, "ERROR", repr(e)[:300], flush=True)
