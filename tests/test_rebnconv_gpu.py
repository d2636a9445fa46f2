"""REBNCONV / RSU4F on the tap-table conv kernels (nnuzoo_amd/rebnconv.py) against the torch modules they replace
(reference classes: /root/reference/nnunetv2/nets/m2net.py:18-30, :769-801): dilated 3x3 conv (1 / 2 / 4 / 8) ->
BatchNorm2d (batch statistics, running-estimate update) -> ReLU, forward and every gradient, under fp16 autocast.
Yardstick = the same torch modules in fp32 (the HIP path must be as close to fp32 as torch's own autocast path, x3)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dil", [1, 2, 4, 8])
@pytest.mark.parametrize("cin,cout,hw", [(64, 64, (32, 32)), (128, 64, (16, 24)), (96, 32, (20, 12))])
def test_dilated_conv_tables_vs_conv2d(hip_lib, dil, cin, cout, hw):
    """the dilated tap tables through conv_box / conv_wgrad: forward, data gradient, weight gradient vs F.conv2d (fp32 on
    the same fp16-rounded operands)"""
    import torch.nn.functional as F
    from nnuzoo_amd import conv_plan as cp, hip_ops as ops
    from nnuzoo_amd.hip_ops import PreparedTable
    N, (H, W) = 2, hw
    g = torch.Generator().manual_seed(dil * 7 + cin)
    x = torch.randn(N, H, W, cin, generator=g).to(torch.float16)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (9 * cin) ** 0.5)
    dy = torch.randn(N, H, W, cout, generator=g).to(torch.float16)
    dims, ks, d3 = (1, H, W), (1, 3, 3), (1, dil, dil)
    pf = PreparedTable(cp.conv_forward(N, dims, cin, cout, ks=ks, dilation=d3))
    pd = PreparedTable(cp.conv_dgrad(N, dims, cin, cout, ks=ks, dilation=d3))
    pw = PreparedTable(cp.conv_wgrad(N, dims, cin, cout, ks=ks, dilation=d3))
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    wf = ops.pack_weight(wd, pf, cin, cout, 9, cin * 9, 1)
    wb = ops.pack_weight(wd, pd, cout, cin, cin * 9, 9, 1)
    y = torch.empty(N, H * W, cout, dtype=torch.float16, device="cuda")
    ops.conv_tap_forward(pf, xd.view(N, H * W, cin), wf, None, y)
    w16 = w.to(torch.float16).float()
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w16.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, padding=dil, dilation=dil)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    ref = yr.detach().permute(0, 2, 3, 1).reshape(N, H * W, cout)
    assert torch.allclose(y.float().cpu(), ref, rtol=3e-3, atol=3e-3 * ref.abs().max().item())
    dx = torch.empty(N, H * W, cin, dtype=torch.float16, device="cuda")
    ops.conv_tap_forward(pd, dyd.view(N, H * W, cout), wb, None, dx)
    rdx = xr.grad.permute(0, 2, 3, 1).reshape(N, H * W, cin)
    assert torch.allclose(dx.float().cpu(), rdx, rtol=3e-3, atol=3e-3 * rdx.abs().max().item())
    ws = torch.empty(ops.conv_tap_wgrad_workspace_floats(pw), device="cuda")
    gw = torch.empty(cout, cin, 3, 3, device="cuda")
    ops.conv_tap_wgrad_to_grad(pw, xd.view(N, H * W, cin), dyd.view(N, H * W, cout), ws, gw, 9, cin * 9, 1)
    assert torch.allclose(gw.cpu(), wr.grad, rtol=2e-3, atol=2e-3 * wr.grad.abs().max().item())


def _run(mod, x, gy, autocast):
    for p in mod.parameters():
        p.grad = None
    xi = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
        y = mod(xi)
    y.backward(gy.to(y.dtype))
    return y.detach().float(), xi.grad.float(), {n: p.grad.float().clone() for n, p in mod.named_parameters()}, \
        {n: b.detach().float().clone() for n, b in mod.named_buffers()}


def test_rsu4f_hip_vs_torch(hip_lib):
    from nnuzoo_amd import rebnconv
    from nnuzoo_amd.nets.common2d import RSU4F
    torch.manual_seed(0)
    hip = RSU4F(128, 64, 128).cuda().train()
    with torch.no_grad():                                  # non-trivial affine parameters / biases
        for n, p in hip.named_parameters():
            if "bn_s1.weight" in n:
                p.copy_(1 + 0.2 * torch.randn_like(p))
            elif n.endswith("bias"):
                p.copy_(0.1 * torch.randn_like(p))
    ref32, ref16 = copy.deepcopy(hip), copy.deepcopy(hip)
    x = torch.randn(2, 128, 32, 32, device="cuda")
    gy = torch.randn(2, 128, 32, 32, device="cuda")
    assert rebnconv.hip_path_ok(hip, x) is False          # outside autocast: torch path
    rebnconv.USE_HIP = False
    y32, dx32, g32, b32 = _run(ref32, x, gy, autocast=False)
    y16, dx16, g16, b16 = _run(ref16, x, gy, autocast=True)
    rebnconv.USE_HIP = True
    with torch.autocast("cuda", dtype=torch.float16):
        assert rebnconv.hip_path_ok(hip, x)
    yh, dxh, gh, bh = _run(hip, x, gy, autocast=True)

    def rel(a, b):
        return ((a - b).norm() / (b.norm() + 1e-12)).item()

    assert rel(yh, y32) < max(5e-3, 3 * rel(y16, y32)), (rel(yh, y32), rel(y16, y32))
    assert rel(dxh, dx32) < max(2e-2, 3 * rel(dx16, dx32)), (rel(dxh, dx32), rel(dx16, dx32))
    for n in g32:
        if n.endswith("conv_s1.bias"):                    # bias in front of batch statistics: exact zero on our side
            assert gh[n].abs().max().item() == 0 and g32[n].abs().max().item() < 1e-3 * max(1.0, g32[n.replace("bias", "weight")].abs().max().item())
            continue
        assert rel(gh[n], g32[n]) < max(3e-2, 3 * rel(g16[n], g32[n])), (n, rel(gh[n], g32[n]), rel(g16[n], g32[n]))
    for n in b32:                                          # running_mean / running_var / num_batches_tracked
        assert torch.allclose(bh[n], b32[n], rtol=5e-3, atol=5e-3), n
    # eval mode (inference): running statistics
    hip.eval(); ref32.eval()
    with torch.no_grad():
        with torch.autocast("cuda", dtype=torch.float16):
            ye = hip(x)
        yr = ref32(x)
    assert rel(ye.float(), yr) < 1e-2


def test_eval_mode_with_trainable_weights_takes_the_torch_path(hip_lib):
    """eval-mode statistics with autograd ON and an input that does not require grad while the weights do (frozen-BN
    fine-tuning, the first RSU4F on raw input) must not take the HIP path (its backward supports batch statistics only):
    forward + backward run through torch and the recorded backend says so"""
    from nnuzoo_amd import rebnconv
    from nnuzoo_amd.nets.common2d import RSU4F
    torch.manual_seed(0)
    m = RSU4F(64, 32, 64).cuda().eval()                                 # channel counts the conv_box path takes
    x = torch.randn(2, 64, 32, 32, device="cuda")                       # requires_grad False
    with torch.autocast("cuda", dtype=torch.float16):
        assert not rebnconv.hip_path_ok(m, x)
        y = m(x)
    assert m.backend == "library"
    y.float().square().mean().backward()                                 # raised RuntimeError inside _RebnConvFn before
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        assert rebnconv.hip_path_ok(m, x)                                # pure inference keeps the HIP path
