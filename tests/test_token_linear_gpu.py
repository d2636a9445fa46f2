"""TokenLinear (nnuzoo_amd/token_linear.py: chunked weight-gradient GEMM for very tall, thin token matrices) against
nn.Linear: fp32 to 1e-5 / 1e-4, fp16 autocast to fp16 rounding of the same quantities."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("autocast", [False, True])
@pytest.mark.parametrize("bias", [False, True])
def test_token_linear_matches_nn_linear(hip_lib, autocast, bias):
    from nnuzoo_amd.token_linear import TokenLinear, _TallLinearFn
    torch.manual_seed(0)
    lin = TokenLinear(16, 64, bias=bias).cuda()
    ref = torch.nn.Linear(16, 64, bias=bias).cuda()
    ref.load_state_dict(lin.state_dict())
    x = torch.randn(2, 256, 256, 16, device="cuda")           # 131 072 tokens: the chunked path
    dy = torch.randn(2, 256, 256, 64, device="cuda")
    res = []
    for m in (lin, ref):
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
            y = m(xi)
        y.backward(dy.to(y.dtype))
        res.append((y.detach().float(), xi.grad.clone(), m.weight.grad.clone(), None if not bias else m.bias.grad.clone()))
    assert res[0][0].dtype == torch.float32 and (lin(x).dtype == torch.float32)
    tol = 2e-3 if autocast else 1e-5
    for a, b in zip(res[0], res[1]):
        if a is None:
            continue
        assert torch.allclose(a, b, rtol=10 * tol, atol=tol * b.abs().max().item()), (a - b).abs().max().item()
    # small inputs take the library path
    assert lin(torch.randn(8, 16, device="cuda")).shape == (8, 64)


@pytest.mark.parametrize("T,K,N,bias", [(5000, 16, 64, True), (4097, 32, 16, False), (70001, 64, 256, True),
                                        (3000, 256, 128, False), (33, 128, 64, True), (2048, 16, 32, False),
                                        (1500, 64, 32, True), (1024, 128, 256, False)])
def test_hip_token_linear_kernels_vs_fp32(hip_lib, T, K, N, bias):
    """csrc/token_linear.hip (forward, input gradient = same kernel with W^T, weight / bias gradient) against fp32 matmuls
    of the SAME fp16-rounded operands: the only differences are the fp32 accumulation order and the final fp16 rounding
    (ragged token counts: tiles of 32 tokens, 64-token staging rounds of the weight gradient)."""
    from nnuzoo_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(T + K)
    x = torch.randn(T, K, generator=g).to(torch.float16)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) if bias else None
    dy = torch.randn(T, N, generator=g).to(torch.float16)
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    bd = b.cuda() if bias else None
    y = torch.empty(T, N, dtype=torch.float16, device="cuda")
    call("nnz_token_linear_forward", ptr(xd), ptr(wd), ptr(bd), ptr(y), T, K, N, 0, stream_ptr())
    w16 = w.to(torch.float16).float()                               # the kernel rounds the master weight to fp16
    ref = x.float() @ w16.t() + (b if bias else 0)
    assert torch.allclose(y.float().cpu(), ref, rtol=2e-3, atol=2e-3 * ref.abs().max().item())
    if hip_lib.nnz_token_linear_supported(N, K):
        dx = torch.empty(T, K, dtype=torch.float16, device="cuda")
        call("nnz_token_linear_forward", ptr(dyd), ptr(wd), None, ptr(dx), T, N, K, 1, stream_ptr())
        rdx = dy.float() @ w16
        assert torch.allclose(dx.float().cpu(), rdx, rtol=2e-3, atol=2e-3 * rdx.abs().max().item())
    if (N // 8) * (K // 8) <= 256:
        buf = torch.zeros(N * K + N, device="cuda")
        call("nnz_token_linear_wgrad", ptr(dyd), ptr(xd), ptr(buf), ptr(buf[N * K:]) if bias else None, T, N, K,
             stream_ptr())
        rdw = dy.float().t() @ x.float()
        got = buf[:N * K].view(N, K).cpu()
        assert torch.allclose(got, rdw, rtol=1e-4, atol=1e-4 * rdw.abs().max().item())
        if bias:
            assert torch.allclose(buf[N * K:].cpu(), dy.float().sum(0), rtol=1e-4, atol=1e-3)


def test_hip_token_linear_module_under_autocast(hip_lib):
    """TokenLinear under fp16 autocast takes the HIP kernels (fp16 output, fp32 parameter gradients) and agrees with
    nn.Linear under the same autocast to fp16 rounding"""
    from nnuzoo_amd.token_linear import TokenLinear
    torch.manual_seed(1)
    lin = TokenLinear(32, 128, bias=True).cuda()
    ref = torch.nn.Linear(32, 128, bias=True).cuda()
    ref.load_state_dict(lin.state_dict())
    x = torch.randn(2, 96, 80, 32, device="cuda").to(torch.float16)
    dy = torch.randn(2, 96, 80, 128, device="cuda").to(torch.float16)
    outs = []
    for m in (lin, ref):
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16):
            y = m(xi)
        assert y.dtype == torch.float16
        y.backward(dy)
        outs.append((y.detach().float(), xi.grad.float(), m.weight.grad.clone(), m.bias.grad.clone()))
        assert m.weight.grad.dtype == torch.float32 and xi.grad.dtype == torch.float16
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=2e-2, atol=3e-3 * b.abs().max().item()), (a - b).abs().max().item()
