import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnuzoo_amd.selective_scan import selective_scan_fn
from nnuzoo_amd.nets.m2net import SS2D
for (B, Di, H, W) in [(2, 32, 128, 128), (2, 32, 64, 64), (2, 64, 32, 32), (2, 256, 4, 4), (2, 256, 2, 2), (2, 256, 1, 1), (2, 128, 8, 8)]:
    K, N, L = 4, 16, H * W
    KD = K * Di
    try:
        u = torch.randn(B, KD, L, device="cuda"); dl = torch.randn(B, KD, L, device="cuda")
        A = -torch.rand(KD, N, device="cuda"); Bm = torch.randn(B, K, N, L, device="cuda"); Cm = torch.randn(B, K, N, L, device="cuda")
        D = torch.randn(KD, device="cuda"); bias = torch.randn(KD, device="cuda")
        y = selective_scan_fn(u, dl, A, Bm, Cm, D, None, bias, True)
        torch.cuda.synchronize()
        print("OK", B, Di, H, W, float(y.abs().mean()), flush=True)
    except Exception as e:
        print("FAIL", B, Di, H, W, repr(e)[:200], flush=True)
SS2D.fused_cross_scan = False
for d_model, H in [(16, 128), (128, 8), (128, 4), (128, 2), (128, 1)]:
    try:
        blk = SS2D(d_model=d_model).cuda()
        with torch.autocast("cuda"):
            y = blk(torch.randn(2, H, H, d_model, device="cuda"))
        torch.cuda.synchronize()
        print("OK block", d_model, H, flush=True)
    except Exception as e:
        print("FAIL block", d_model, H, repr(e)[:200], flush=True)
