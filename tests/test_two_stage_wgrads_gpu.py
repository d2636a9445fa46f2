"""Two-stage (partial blocks + fixed-order fold) weight gradients of the token Linear, the SS2D x_proj and the SS2D depthwise conv +
SiLU: equal to float64 references like the atomic forms, and BIT-IDENTICAL from call to call (the atomic forms are not: their
low bits follow the arrival order of the workgroups) - csrc/common.hpp fold_partials, round 4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _call(name, *a):
    from nnuzoo_amd._lib import call
    call(name, *a)


def test_token_linear_wgrad_two_stage(hip_lib):
    from nnuzoo_amd._lib import load, ptr, stream_ptr
    g = torch.Generator().manual_seed(3)
    for T, N, K in ((262144, 32, 16), (32768, 128, 128), (4096 + 64, 64, 256), (100, 8, 8)):
        dy = torch.randn(T, N, generator=g).to(torch.float16).to(DEV)
        x = torch.randn(T, K, generator=g).to(torch.float16).to(DEV)
        ref_w = dy.double().t() @ x.double()
        ref_b = dy.double().sum(0)
        nws = int(load().nnz_token_linear_wgrad_workspace_floats(T, N, K))
        outs = []
        for rep in range(3):
            ws = torch.full((nws,), float("nan"), device=DEV)        # the kernel must write every partial it later folds
            buf = torch.full((N * K + N,), float("nan"), device=DEV)
            _call("nnz_token_linear_wgrad_ws", ptr(dy), ptr(x), ptr(buf[:N * K]), ptr(buf[N * K:]), ptr(ws), nws, T, N, K, stream_ptr())
            torch.cuda.synchronize()
            outs.append(buf.clone())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (T, N, K)
        dw, db = outs[0][:N * K].view(N, K), outs[0][N * K:]
        scale = ref_w.abs().max().item()
        assert (dw.double() - ref_w).abs().max().item() <= 2e-5 * scale + 1e-3, (T, N, K)
        assert (db.double() - ref_b).abs().max().item() <= 2e-5 * ref_b.abs().max().item() + 1e-3
        # the atomic form agrees to summation-order noise
        buf2 = torch.zeros(N * K + N, device=DEV)
        _call("nnz_token_linear_wgrad", ptr(dy), ptr(x), ptr(buf2[:N * K]), ptr(buf2[N * K:]), T, N, K, stream_ptr())
        torch.cuda.synchronize()
        assert torch.allclose(buf2, outs[0], rtol=1e-4, atol=1e-4 * scale)


def test_xproj_backward_w_two_stage(hip_lib):
    from nnuzoo_amd._lib import load, ptr, stream_ptr
    g = torch.Generator().manual_seed(4)
    for B, Di, Cp, L in ((2, 32, 17, 512 * 512), (2, 128, 20, 64 * 64), (1, 64, 18, 128 * 64)):
        C2 = 2 * Cp
        dP = torch.randn(2, B, C2, L, generator=g).to(DEV)
        x2 = torch.randn(2, B, Di, L, generator=g).to(DEV)
        ref = torch.einsum("sbcl,sbdl->scd", dP.double(), x2.double())          # [2][C2][Di]
        ref = ref.view(2, 2, Cp, Di).transpose(0, 1).reshape(4, Cp, Di)           # the module's [K = 4][Cp][Di] layout
        nws = int(load().nnz_ss2d_xproj_backward_w_workspace_floats(B, Di, C2, L))
        outs = []
        for rep in range(3):
            ws = torch.full((nws,), float("nan"), device=DEV)
            dW = torch.full((4, Cp, Di), float("nan"), device=DEV)
            _call("nnz_ss2d_xproj_backward_w_ws", ptr(dP), ptr(x2), ptr(dW), ptr(ws), nws, B, Di, C2, L, Cp, stream_ptr())
            torch.cuda.synchronize()
            outs.append(dW.clone())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (B, Di, Cp, L)
        assert (outs[0].double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-3


def test_dwconv_silu_backward_two_stage(hip_lib):
    from nnuzoo_amd._lib import load, ptr, stream_ptr
    g = torch.Generator().manual_seed(5)
    for B, D, H, W in ((2, 32, 512, 512), (2, 64, 100, 36), (1, 256, 16, 16)):
        x = torch.randn(B, H, W, D, generator=g).to(DEV)
        w = (0.3 * torch.randn(D, 1, 3, 3, generator=g)).to(DEV)
        b = (0.1 * torch.randn(D, generator=g)).to(DEV)
        dx2 = torch.randn(2, B, D, H * W, generator=g).to(DEV)
        # float64 reference of x2 = [silu(conv(x)) row-major, the same column-major]
        xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
        wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
        y = torch.nn.functional.silu(torch.nn.functional.conv2d(xr, wr, br, padding=1, groups=D))
        x2r = torch.stack([y.flatten(2), y.transpose(2, 3).flatten(2)])
        x2r.backward(dx2.double())
        nws = int(load().nnz_ss2d_dwconv_silu_backward_workspace_floats(B, D, H, W))
        outs = []
        for rep in range(3):
            ws = torch.full((nws,), float("nan"), device=DEV)
            dwb = torch.full((D * 10,), float("nan"), device=DEV)
            dx = torch.empty(B, H, W, D, device=DEV)
            _call("nnz_ss2d_dwconv_silu_backward_ws", ptr(x), 0, D, ptr(w.view(D, 9).contiguous()), ptr(b), ptr(dx2), ptr(dx),
                  ptr(dwb[:D * 9]), ptr(dwb[D * 9:]), ptr(ws), nws, B, D, H, W, stream_ptr())
            torch.cuda.synchronize()
            outs.append(dwb.clone())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (B, D, H, W)
        dw, db = outs[0][:D * 9].view(D, 1, 3, 3), outs[0][D * 9:]
        assert (dw.double() - wr.grad).abs().max().item() <= 5e-5 * wr.grad.abs().max().item() + 1e-3
        assert (db.double() - br.grad).abs().max().item() <= 5e-5 * br.grad.abs().max().item() + 1e-3
        assert (dx.double().permute(0, 3, 1, 2) - xr.grad).abs().max().item() <= 1e-4 * xr.grad.abs().max().item()


def test_grouped_token_linear_weight_gradients(hip_lib):
    """round 5: inside deferred_wgrads() the fp16 token Linears queue their weight gradients; ONE grouped launch + ONE fold launch
    (csrc/token_linear.hip tl_wgrad_group_kernel) computes them: equal to float64, bit-identical from pass to pass, and equal to the
    per-layer two-stage launches up to the fold order"""
    from nnuzoo_amd import token_linear as TLm
    from nnuzoo_amd.token_linear import TokenLinear, deferred_wgrads
    torch.manual_seed(5)
    shapes = [(4096, 32, 64), (70000, 16, 32), (1024, 128, 128), (2048, 64, 16), (1500, 128, 64)]   # T, K, N
    lins = [TokenLinear(K, N, bias=(i % 2 == 0)).to(DEV) for i, (T, K, N) in enumerate(shapes)]
    xs = [torch.randn(T, K, device=DEV) for T, K, N in shapes]

    def run(deferred):
        for l in lins:
            l.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = sum((l(x).float() * (1 + i)).square().mean() for i, (l, x) in enumerate(zip(lins, xs)))
        if deferred:
            with deferred_wgrads():
                loss.backward()
        else:
            loss.backward()
        torch.cuda.synchronize()
        assert all(l.backend == "hip-f16" for l in lins)
        return [p.grad.clone() for l in lins for p in l.parameters()]

    a, b = run(True), run(True)
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    old = TLm.TWO_STAGE
    try:
        TLm.TWO_STAGE = True
        c = run(False)
    finally:
        TLm.TWO_STAGE = old
    for u, v in zip(a, c):
        assert torch.allclose(u, v, rtol=2e-5, atol=2e-5 * v.abs().max().item())
    # float64 reference of the first layer's weight gradient from the same fp16 operands
    l, x = lins[0], xs[0]
    xh = x.to(torch.float16)
    y = (xh.double() @ l.weight.to(torch.float16).double().t() + l.bias.to(torch.float16).double()).to(torch.float16).float()
    dy = (2 * y / y.numel()).to(torch.float16)
    ref = dy.double().t() @ xh.double()
    assert (a[0].double() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
