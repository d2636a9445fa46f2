"""HIP LayerNorm (csrc/layer_norm.hip, nnuzoo_amd.layer_norm.LayerNorm) against torch's F.layer_norm in fp32: forward
rtol 1e-5, input gradient rtol 1e-4, dgamma / dbeta rtol 1e-4 (atomic fp32 sums over workgroups).  Shapes are the
token-major activations of the VSS / Swin blocks (C = 16 .. 1024, non-power-of-two Swin widths)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 64, 64, 16), (2, 33, 17, 32), (3, 50, 64), (1, 29, 31, 96), (2, 16, 16, 128),
                                   (5, 7, 192), (4, 9, 256), (3, 11, 384), (2, 5, 512), (7, 1024), (3, 2048)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_layer_norm_matches_torch(hip_lib, shape, dtype):
    from nnuzoo_amd.layer_norm import LayerNorm
    g = torch.Generator().manual_seed(5)
    C = shape[-1]
    x = (torch.randn(shape, generator=g) * 2 + 0.5).to(dtype).cuda()
    ln = LayerNorm(C, eps=1e-5).cuda()
    with torch.no_grad():
        ln.weight.copy_(torch.randn(C, generator=g) * 0.3 + 1)
        ln.bias.copy_(torch.randn(C, generator=g) * 0.2)
    dy = torch.randn(shape, generator=g).cuda()
    xa = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16, enabled=dtype == torch.float16):
        y = ln(xa)
    assert y.dtype == torch.float32
    y.backward(dy)
    xr = x.float().clone().requires_grad_(True)
    w, b = ln.weight.detach().clone().requires_grad_(True), ln.bias.detach().clone().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), w, b, 1e-5)
    yr.backward(dy)
    assert torch.allclose(y, yr, rtol=1e-5, atol=1e-5), (y - yr).abs().max().item()
    assert xa.grad.dtype == dtype
    tol = 1e-4 if dtype == torch.float32 else 2e-3
    assert torch.allclose(xa.grad.float(), xr.grad, rtol=tol, atol=tol * xr.grad.abs().max().item())
    assert torch.allclose(ln.weight.grad, w.grad, rtol=1e-4, atol=1e-4 * w.grad.abs().max().item())
    assert torch.allclose(ln.bias.grad, b.grad, rtol=1e-4, atol=1e-4 * b.grad.abs().max().item())


def test_layer_norm_large_rows_and_no_affine(hip_lib):
    from nnuzoo_amd.layer_norm import LayerNorm, layer_norm
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 512, 512, 16, generator=g).cuda().requires_grad_(True)     # more rows than the grid covers
    ln = LayerNorm(16).cuda()
    y = ln(x)
    y.sum().backward()
    yr = F.layer_norm(x.detach(), (16,), ln.weight.detach(), ln.bias.detach())
    assert torch.allclose(y, yr, rtol=1e-5, atol=1e-5)
    assert torch.allclose(ln.bias.grad, torch.full((16,), 2.0 * 512 * 512, device="cuda"))
    z = layer_norm(x.detach(), None, None, 1e-6)
    assert torch.allclose(z, F.layer_norm(x.detach(), (16,), None, None, 1e-6), rtol=1e-5, atol=1e-5)


def test_layer_norm_refuses_cpu():
    from nnuzoo_amd.layer_norm import LayerNorm
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        LayerNorm(16)(torch.zeros(2, 16))


@pytest.mark.parametrize("C,zdtype", [(32, torch.float16), (64, torch.float32), (256, torch.float16), (96, torch.float32)])
def test_layer_norm_gate_matches_torch(hip_lib, C, zdtype):
    """LN(y) * silu(z) with z a strided half of a wider tensor (the SS2D in_proj output), against torch ops in fp32"""
    from nnuzoo_amd.layer_norm import layer_norm_gate
    g = torch.Generator().manual_seed(C)
    B, H, W = 2, 11, 13
    y = torch.randn(B, H, W, C, generator=g).cuda().requires_grad_(True)
    xz = torch.randn(B, H, W, 2 * C, generator=g).to(zdtype).cuda().requires_grad_(True)
    w = (torch.randn(C, generator=g) * 0.3 + 1).cuda().requires_grad_(True)
    b = (torch.randn(C, generator=g) * 0.2).cuda().requires_grad_(True)
    dout = torch.randn(B, H, W, C, generator=g).cuda()
    z = xz.chunk(2, dim=-1)[1]
    out = layer_norm_gate(y, z, w, b, 1e-5)
    out.backward(dout)
    got = [out.detach(), y.grad.clone(), xz.grad.float().clone(), w.grad.clone(), b.grad.clone()]
    for t in (y, xz, w, b):
        t.grad = None
    z = xz.float().chunk(2, dim=-1)[1]
    ref = F.layer_norm(y, (C,), w, b, 1e-5) * F.silu(z)
    ref.backward(dout)
    want = [ref.detach(), y.grad, xz.grad.float(), w.grad, b.grad]
    tols = [1e-5, 1e-4, 1e-4 if zdtype == torch.float32 else 2e-3, 1e-4, 1e-4]
    for a, r, tol in zip(got, want, tols):
        assert torch.allclose(a, r, rtol=tol, atol=tol * r.abs().max().item()), (a - r).abs().max().item()


def test_layer_norm_half_output_for_linear_consumers(hip_lib):
    """feeds_linear: under fp16 autocast the kernel writes fp16 (what the consuming Linear's own cast would produce) and
    reads the fp16 gradient that Linear returns; outside autocast nothing changes"""
    from nnuzoo_amd.layer_norm import LayerNorm, layer_norm_gate
    g = torch.Generator().manual_seed(9)
    C = 64
    x = torch.randn(2, 9, 7, C, generator=g).cuda()
    ln = LayerNorm(C).cuda()
    ln.feeds_linear = True
    lin = torch.nn.Linear(C, 24).cuda()
    assert ln(x).dtype == torch.float32
    xa = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        y = ln(xa)
        out = lin(y)
    assert y.dtype == torch.float16
    out.float().sum().backward()
    xr = x.clone().requires_grad_(True)
    w, b = ln.weight.detach().clone().requires_grad_(True), ln.bias.detach().clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        yr = F.layer_norm(xr, (C,), w, b, 1e-5)
        outr = lin(yr)
    outr.float().sum().backward()
    assert torch.allclose(y.float(), yr.half().float(), rtol=2e-3, atol=2e-3)
    assert torch.allclose(out.float(), outr.float(), rtol=1e-2, atol=1e-2)
    assert torch.allclose(xa.grad, xr.grad, rtol=1e-2, atol=1e-2 * xr.grad.abs().max().item())
    assert torch.allclose(ln.weight.grad, w.grad, rtol=1e-2, atol=1e-2 * w.grad.abs().max().item())
    # gated variant
    z = torch.randn(2, 9, 7, C, generator=g).cuda().half()
    with torch.autocast("cuda", dtype=torch.float16):
        yg = layer_norm_gate(x, z, ln.weight, ln.bias, 1e-5, feeds_linear=True)
    assert yg.dtype == torch.float16
    ref = F.layer_norm(x, (C,), ln.weight, ln.bias, 1e-5) * F.silu(z.float())
    assert torch.allclose(yg.float(), ref, rtol=2e-3, atol=2e-3)


def test_affine_gradients_are_bit_identical_run_to_run(hip_lib):
    """round 3: dgamma / dbeta are fixed-point cross-workgroup sums (csrc/common.hpp FxAcc) written by the launch's last
    workgroup - no float atomics: two backward passes give the same bits, and the shared scratch is left zero"""
    import torch
    from nnuzoo_amd.hip_ops import det_scratch
    from nnuzoo_amd.layer_norm import LayerNorm
    torch.manual_seed(0)
    for C, rows in [(96, 35378), (768, 882), (16, 262144)]:
        ln = LayerNorm(C).cuda()
        x = torch.randn(rows, C, device="cuda", requires_grad=True)
        dy = torch.randn(rows, C, device="cuda")
        outs = []
        for _ in range(2):
            gx, gw, gb = torch.autograd.grad(ln(x), [x, ln.weight, ln.bias], dy)
            outs.append((gw.clone(), gb.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        xr = x.detach().double()
        xh = (xr - xr.mean(1, keepdim=True)) / torch.sqrt(xr.var(1, unbiased=False, keepdim=True) + ln.eps)
        assert torch.allclose(outs[0][0].double(), (dy.double() * xh).sum(0), rtol=1e-4, atol=1e-3)
        assert torch.allclose(outs[0][1].double(), dy.double().sum(0), rtol=1e-4, atol=1e-3)
    sc = det_scratch(torch.device("cuda", 0), 0)
    assert int(sc.acc.abs().max()) == 0 and int(sc.counter.abs().max()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 14, 14, 32), (3, 35, 35, 96), (1, 7, 7, 768), (5, 9, 64)])
def test_layer_norm_skip_sums_both_gradients_in_the_kernel(hip_lib, shape):
    """layer_norm_skip: (norm(x), x) whose backward adds the residual stream's gradient inside ln_bwd_kernel
    (nnz_layer_norm_backward_det_res) - against x + f(norm(x)) written with the plain module and autograd's own add;
    1e-6: the in-kernel sum may contract into an fma.  Also: only one of the two outputs used."""
    from nnuzoo_amd.layer_norm import LayerNorm, layer_norm_skip
    torch.manual_seed(0)
    C = shape[-1]
    norm = LayerNorm(C).cuda()
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
    x0 = torch.randn(*shape, device="cuda")
    w = torch.randn(C, C, device="cuda") / C ** 0.5
    dout = torch.randn(*shape, device="cuda")
    res = []
    for fused in (True, False):
        x = x0.clone().requires_grad_(True)
        h = x * 1.0                                    # a non-leaf, like the block input
        n, s = layer_norm_skip(norm, h) if fused else (norm(h), h)
        if fused:
            assert type(n.grad_fn).__name__.startswith("_LayerNormSkipFn")
        out = s + torch.tanh(n @ w)
        gx, gw, gb = torch.autograd.grad(out, [x, norm.weight, norm.bias], dout)
        res.append((out.detach(), gx, gw, gb))
    assert torch.equal(res[0][0], res[1][0])
    for u, v, name in zip(res[0][1:], res[1][1:], ("dx", "dgamma", "dbeta")):
        assert torch.allclose(u, v, rtol=1e-5, atol=1e-6 * max(1.0, v.abs().max().item())), (name, (u - v).abs().max().item())
    # one output only: the skip alone passes the gradient through, the norm alone is the plain backward
    x = x0.clone().requires_grad_(True)
    n, s = layer_norm_skip(norm, x * 1.0)
    (g1,) = torch.autograd.grad(s, x, dout, retain_graph=True)
    assert torch.equal(g1, dout)
    (g2,) = torch.autograd.grad(n, x, dout)
    x2 = x0.clone().requires_grad_(True)
    (g3,) = torch.autograd.grad(norm(x2 * 1.0), x2, dout)
    assert torch.equal(g2, g3)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,feeds", [((2, 16, 16, 32), True), ((2, 64, 64, 16), True), ((2, 32, 32, 256), False), ((2, 128, 128, 16), True)])
@pytest.mark.parametrize("xdtype", [torch.float16, torch.float32])
def test_layer_norm_skip_under_fp16_autocast(hip_lib, shape, feeds, xdtype):
    """the VSS block's use (m2net.py:530 `input + drop_path(self_attention(ln_1(input)))`): an fp16 or fp32 residual stream inside an
    fp16-autocast region, fp16 rows out when the norm feeds an autocast Linear.  Against the un-fused form (the plain module + autograd's
    own add): identical forward bits; dx to the rounding of the stream's type (the fused sum is formed in fp32 and rounded once)."""
    from nnuzoo_amd.layer_norm import LayerNorm, layer_norm_skip
    torch.manual_seed(1)
    C = shape[-1]
    norm = LayerNorm(C).cuda()
    norm.feeds_linear = feeds
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
    x0 = torch.randn(*shape, device="cuda").to(xdtype)
    w = (torch.randn(C, C, device="cuda") / C ** 0.5)
    dout = torch.randn(*shape, device="cuda").to(xdtype)
    res = []
    with torch.autocast("cuda", dtype=torch.float16):
        for fused in (True, False):
            x = x0.clone().requires_grad_(True)
            h = x * 1.0
            n, s = layer_norm_skip(norm, h) if fused else (norm(h), h)
            if fused:
                assert type(n.grad_fn).__name__.startswith("_LayerNormSkipFn")
            assert n.dtype == (torch.float16 if feeds else torch.float32)
            out = s + torch.tanh(torch.nn.functional.linear(n, w)).to(s.dtype)
            gx, gw, gb = torch.autograd.grad(out, [x, norm.weight, norm.bias], dout)
            res.append((out.detach(), gx, gw, gb))
    assert torch.equal(res[0][0], res[1][0])
    eps = 2.0 ** -10 if xdtype == torch.float16 else 1e-6
    for u, v, name in zip(res[0][1:], res[1][1:], ("dx", "dgamma", "dbeta")):
        tol = (2 * eps if name == "dx" else 1e-5) * max(1.0, v.float().abs().max().item())
        assert (u.float() - v.float()).abs().max().item() <= tol, (name, (u.float() - v.float()).abs().max().item())
