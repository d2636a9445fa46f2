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
