"""csrc/upsample.hip: bilinear up-sampling (`_upsample_like`, /root/reference/nnunetv2/nets/m2net.py:33-36) and its adjoint against
torch's own F.interpolate / autograd in fp32 (the plain PyTorch formulation of the op)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [((2, 2, 16, 16), (512, 512)), ((2, 3, 32, 24), (64, 48)), ((1, 2, 7, 5), (33, 31)), ((2, 2, 64, 64), (128, 128)),
         ((1, 4, 9, 11), (9, 11)), ((2, 1, 40, 56), (20, 28)), ((1, 2, 1, 1), (8, 8)), ((1, 1, 3, 200), (24, 1600))]


@pytest.mark.parametrize("shape,size", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_forward_and_adjoint_match_torch(hip_lib, shape, size, dtype):
    from nnuzoo_amd.nets.common2d import _BilinearUpFn
    torch.manual_seed(sum(shape) + size[0])
    x = torch.randn(shape, device="cuda").to(dtype).requires_grad_(True)
    g = torch.randn(shape[0], shape[1], *size, device="cuda").to(dtype)
    y = _BilinearUpFn.apply(x, size)
    assert y.dtype == dtype and y.shape == g.shape
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    yr = F.interpolate(xr, size=size, mode="bilinear", align_corners=False)
    yr.backward(g.float())
    eps = 2.0 ** -10 if dtype == torch.float16 else 4e-6
    assert (y.float() - yr).abs().max().item() <= 2 * eps * max(1.0, yr.abs().max().item())
    assert (x.grad.float() - xr.grad).abs().max().item() <= 2 * eps * max(1.0, xr.grad.abs().max().item())


def test_adjoint_identity_and_repeatability(hip_lib):
    """<up(x), g> == <x, up^T(g)> to fp32 rounding, and two backward launches give the same bits (fixed-order gather, no atomics)"""
    from nnuzoo_amd.nets.common2d import _BilinearUpFn
    torch.manual_seed(0)
    x = torch.randn(2, 2, 16, 16, device="cuda", requires_grad=True)
    g = torch.randn(2, 2, 512, 512, device="cuda")
    y = _BilinearUpFn.apply(x, (512, 512))
    (gx,) = torch.autograd.grad(y, x, g, retain_graph=True)
    (gx2,) = torch.autograd.grad(y, x, g)
    assert torch.equal(gx, gx2)
    lhs, rhs = (y.double() * g.double()).sum().item(), (x.double() * gx.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0)
