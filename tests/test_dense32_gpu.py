"""fp32 token-major Linear layers on the fp32 matrix cores (csrc/dense32.hip, through the C-ABI) against float64 on the
same inputs: forward (+ bias, + exact GELU with both outputs), input gradient (+ GELU'), weight + bias gradient (token
splits folded in a fixed order: bit-identical run to run).  These replace torch's fp32 F.linear / GELU inside the Swin and
ViT blocks of the reference (swt2net.py:496-515 Mlp, :584-619 qkv / proj; trainer without autocast,
nnUNetTrainerSwT2Net.py:112-130).  Tolerance: fp32 accumulation in a different order than a float64 reference ->
2e-6 * sqrt(contraction length) of the output scale (random-sign sums)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from nnuzoo_amd import _lib
from nnuzoo_amd._lib import call, ptr, stream_ptr

DEV = "cuda"
SHAPES = [(35378, 96, 288), (882, 768, 3072), (882, 3072, 768), (100, 64, 36), (9800, 192, 192), (64, 4, 4),
          (2450, 384, 1536), (333, 100, 260), (70000, 32, 32)]


def _close(got, ref, contraction, what):
    ref = ref.to(got.device)
    scale = ref.abs().max().item() + 1e-30
    err = (got.double() - ref).abs().max().item()
    assert err <= 2e-6 * math.sqrt(contraction) * scale + 1e-30, (what, err, scale)


@pytest.mark.parametrize("T,K,N", SHAPES)
def test_forward_dgrad_wgrad(hip_lib, T, K, N):
    g = torch.Generator().manual_seed(T + K)
    x = torch.randn(T, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    dy = torch.randn(T, N, generator=g).to(DEV)
    x64, W64, b64, dy64 = x.double(), W.double(), b.double(), dy.double()
    # forward, bias
    y = torch.full((T, N), float("nan"), device=DEV)
    call("nnz_dense32_forward", ptr(x), ptr(W), ptr(b), ptr(y), None, T, K, N, 0, stream_ptr())
    _close(y, x64 @ W64.t() + b64, K, "y")
    # forward, no bias, GELU: both outputs
    h, a = torch.full((T, N), float("nan"), device=DEV), torch.full((T, N), float("nan"), device=DEV)
    call("nnz_dense32_forward", ptr(x), ptr(W), None, ptr(h), ptr(a), T, K, N, 1, stream_ptr())
    h64 = x64 @ W64.t()
    _close(h, h64, K, "h")
    _close(a, torch.nn.functional.gelu(h64), K, "gelu(h)")
    # input gradient, plain and times GELU'(pre-activation of the layer below)
    dx = torch.full((T, K), float("nan"), device=DEV)
    call("nnz_dense32_dgrad", ptr(dy), ptr(W), None, ptr(dx), T, K, N, stream_ptr())
    _close(dx, dy64 @ W64, N, "dx")
    pre = torch.randn(T, K, generator=g).to(DEV)
    call("nnz_dense32_dgrad", ptr(dy), ptr(W), ptr(pre), ptr(dx), T, K, N, stream_ptr())
    p64 = pre.double().requires_grad_(True)
    (gp,) = torch.autograd.grad(torch.nn.functional.gelu(p64), p64, torch.ones_like(p64))
    _close(dx, (dy64 @ W64) * gp, N, "dx * gelu'")
    # weight + bias gradient; twice: bit-identical
    ws = torch.empty(int(_lib.load().nnz_dense32_wgrad_workspace_floats(T, K, N)), device=DEV)
    outs = []
    for _ in range(2):
        dW, db = torch.full((N, K), float("nan"), device=DEV), torch.full((N,), float("nan"), device=DEV)
        call("nnz_dense32_wgrad", ptr(dy), ptr(x), ptr(dW), ptr(db), ptr(ws), T, K, N, stream_ptr())
        outs.append((dW, db))
    _close(outs[0][0], dy64.t() @ x64, T, "dW")
    _close(outs[0][1], dy64.sum(0), T, "db")
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    dW2 = torch.full((N, K), float("nan"), device=DEV)
    call("nnz_dense32_wgrad", ptr(dy), ptr(x), ptr(dW2), None, ptr(ws), T, K, N, stream_ptr())     # no bias gradient
    assert torch.equal(dW2, outs[0][0])


def test_token_linear_fp32_module_matches_torch(hip_lib):
    """TokenLinear in an fp32 step (no autocast) takes the hip-f32 backend and equals F.linear's values and gradients"""
    from nnuzoo_amd.token_linear import TokenLinear
    torch.manual_seed(0)
    m = TokenLinear(96, 288).to(DEV)
    x = torch.randn(2, 35, 49, 96, device=DEV, requires_grad=True)
    dy = torch.randn(2, 35, 49, 288, device=DEV)
    y = m(x)
    assert m.backend == "hip-f32"
    gx, gw, gb = torch.autograd.grad(y, [x, m.weight, m.bias], dy)
    xr = x.detach().double().requires_grad_(True)
    wr, br = m.weight.detach().double().requires_grad_(True), m.bias.detach().double().requires_grad_(True)
    yr = torch.nn.functional.linear(xr, wr, br)
    rx, rw, rb = torch.autograd.grad(yr, [xr, wr, br], dy.double())
    _close(y, yr.detach(), 96, "y")
    _close(gx, rx, 288, "dx")
    _close(gw, rw, 3430, "dW")
    _close(gb, rb, 3430, "db")
    m2 = TokenLinear(64, 128).to(DEV)
    with torch.autocast("cuda", dtype=torch.float16):          # autocast steps keep their fp16 kernels
        m2(torch.randn(2, 35, 49, 64, device=DEV))
    assert m2.backend == "hip-f16"


def test_fused_mlp_matches_torch(hip_lib):
    from nnuzoo_amd.nets.swt2net import Mlp
    torch.manual_seed(1)
    m = Mlp(192, 768).to(DEV)
    x = torch.randn(2, 70, 70, 192, device=DEV, requires_grad=True)
    dy = torch.randn(2, 70, 70, 192, device=DEV)
    y = m(x)
    assert type(y.grad_fn).__name__.startswith("_Dense32MlpFn")
    params = [m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias]
    grads = torch.autograd.grad(y, [x] + params, dy)
    xr = x.detach().double().requires_grad_(True)
    pr = [p.detach().double().requires_grad_(True) for p in params]
    yr = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(xr, pr[0], pr[1])), pr[2], pr[3])
    ref = torch.autograd.grad(yr, [xr] + pr, dy.double())
    _close(y, yr.detach(), 768, "y")
    for name, a, r in zip(["dx", "dW1", "db1", "dW2", "db2"], grads, ref):
        _close(a, r, 9800, name)


def test_grouped_deferred_weight_gradients_equal_per_layer_launches(hip_lib):
    """inside deferred_wgrads() the Linear / Mlp nodes queue their weight gradients and ONE grouped launch (+ one fold)
    produces them: bit-identical to the per-layer launches, .grad assigned (and accumulated on a second pass)"""
    from nnuzoo_amd.nets.swt2net import Mlp
    from nnuzoo_amd.token_linear import TokenLinear, deferred_wgrads
    torch.manual_seed(2)
    layers = [TokenLinear(96, 288), Mlp(192, 768), TokenLinear(768, 3072, bias=False), TokenLinear(100, 36),
              TokenLinear(3072, 768), Mlp(96, 384)]
    toks = [35378, 9800, 882, 70, 882, 333]
    net = torch.nn.ModuleList(layers).to(DEV)
    xs = [torch.randn(t, (m.in_features if isinstance(m, TokenLinear) else m.fc1.in_features), device=DEV) for m, t in
          zip(net, toks)]

    def run(deferred: bool):
        for p in net.parameters():
            p.grad = None
        loss = sum((m(x) * (1 + i)).square().mean() for i, (m, x) in enumerate(zip(net, xs)))
        if deferred:
            with deferred_wgrads():
                loss.backward()
        else:
            loss.backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in net.parameters()]

    ref = run(False)
    got = run(True)
    assert len(ref) == len(got) == len(list(net.parameters()))
    assert all(torch.equal(a, b) for a, b in zip(ref, got))
    # accumulation into existing .grad
    loss = sum(m(x).mean() for m, x in zip(net, xs))
    with deferred_wgrads():
        loss.backward()
    extra = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    sum(m(x).mean() for m, x in zip(net, xs)).backward()
    assert all(torch.allclose(e, g + p.grad, rtol=1e-6, atol=1e-7) for e, g, p in zip(extra, got, net.parameters()))


@pytest.mark.parametrize("T,K,N", [(128, 768, 192), (512, 384, 96), (2048, 192, 768), (100, 96, 40), (4096, 1536, 384)])
def test_half_activations_in_and_out(hip_lib, T, K, N):
    """nnz_dense32_forward_h16 / _dgrad_h16 / the fp16 grouped weight-gradient record (round 5): the token Linears of the autocast
    nets that the fp16 token kernel does not take.  Against float64 on the SAME fp16-rounded activations and fp32 weights: the
    product is exact up to fp32 accumulation, the result is rounded once to fp16 (2^-11 relative) - tighter than
    torch.autocast's F.linear, which also rounds W to fp16 (nnUNetTrainer.py:1128-1139)."""
    g = torch.Generator().manual_seed(T + N)
    x = torch.randn(T, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    dy = torch.randn(T, N, generator=g).half().to(DEV)
    lib = _lib.load()
    y = torch.empty(T, N, dtype=torch.float16, device=DEV)
    ws = torch.empty(max(1, int(lib.nnz_dense32_splitk_workspace_floats(T, K, N))), device=DEV)
    call("nnz_dense32_forward_h16", ptr(x), ptr(W), ptr(b), ptr(y), T, K, N, ptr(ws), stream_ptr())
    ref = x.double() @ W.double().t() + b.double()
    err = (y.double() - ref).abs().max().item()
    assert err <= (2.0 ** -11 + 2e-6 * math.sqrt(K)) * ref.abs().max().item(), (err, ref.abs().max().item())
    dx = torch.empty(T, K, dtype=torch.float16, device=DEV)
    ws = torch.empty(max(1, int(lib.nnz_dense32_splitk_workspace_floats(T, N, K))), device=DEV)
    call("nnz_dense32_dgrad_h16", ptr(dy), ptr(W), ptr(dx), T, K, N, ptr(ws), stream_ptr())
    ref = dy.double() @ W.double()
    err = (dx.double() - ref).abs().max().item()
    assert err <= (2.0 ** -11 + 2e-6 * math.sqrt(N)) * ref.abs().max().item(), (err, ref.abs().max().item())
    # the module path under autocast: fp16 rows in, fp16 rows out, weight gradient through the grouped launch
    from nnuzoo_amd.token_linear import TokenLinear, deferred_wgrads
    m = TokenLinear(K, N).to(DEV)
    with torch.no_grad():
        m.weight.copy_(W)
        m.bias.copy_(b)
    xr = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        out = m(xr.view(1, T, K))
    if m.backend == "hip-f32":            # (shapes the fp16 token kernel takes keep that kernel: nothing to compare here)
        assert out.dtype == torch.float16
        with deferred_wgrads():
            out.backward(dy.view(1, T, N))
        torch.cuda.synchronize()
        assert torch.equal(out.view(T, N), y)
        assert torch.equal(xr.grad, dx)
        _close(m.weight.grad, dy.double().t() @ x.double(), T, "dW")
        _close(m.bias.grad, dy.double().sum(0), T, "db")
