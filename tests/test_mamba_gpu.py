"""1-D Mamba block on the HIP operators against fixtures produced by the reference's vendored module
(nets/seg_mamba/mamba_simple.py run on CPU, tools/make_golden.py gen_mamba): forward, input gradient and every
parameter gradient for bimamba_type none / v2 / v3; plus the two element-wise kernels against torch formulas.
Tolerance: fp32 kernels, 3e-4 of the tensor's scale (the scan's exp / softplus use fast intrinsics)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "mamba_block.npz"))


def close(got, ref, tol=3e-4, what=""):
    ref = torch.as_tensor(ref)
    scale = ref.abs().max().item() + 1e-12
    err = (got.detach().cpu().float() - ref).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("tag", ["none", "v2", "v3"])
def test_mamba_module_matches_reference(hip_lib, tag):
    from nnuzoo_amd.nets.mamba_simple import Mamba
    d_model, L, ns = (int(v) for v in GOLD[f"{tag}_cfg"])
    m = Mamba(d_model, bimamba_type=tag, nslices=ns)
    names = [n for n, _ in m.named_parameters()]
    assert names == [k[len(tag) + 3:] for k in GOLD.files if k.startswith(f"{tag}_p_")]   # same parameters, same order
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.from_numpy(GOLD[f"{tag}_p_{n}"]))
    m = m.cuda()
    x = torch.from_numpy(GOLD[f"{tag}_x"]).cuda().requires_grad_(True)
    G = torch.from_numpy(GOLD[f"{tag}_G"]).cuda()
    y = m(x)
    close(y, GOLD[f"{tag}_y"], what="y")
    (y * G).sum().backward()
    close(x.grad, GOLD[f"{tag}_dx"], what="dx")
    used = 0
    for n, p in m.named_parameters():
        key = f"{tag}_g_{n}"
        if key in GOLD.files:
            close(p.grad, GOLD[key], tol=1e-3, what=n)
            used += 1
        else:
            assert p.grad is None or p.grad.abs().max().item() == 0, n   # branch not used by this bimamba_type
    assert used == {"none": 9, "v2": 16, "v3": 23}[tag]


@pytest.mark.parametrize("Bn,D,L,W", [(2, 8, 37, 4), (1, 5, 4100, 4), (3, 16, 9, 3), (1, 4, 300, 8)])
def test_causal_conv1d_silu_kernel(hip_lib, Bn, D, L, W):
    from nnuzoo_amd.mamba_block import causal_conv1d_fn
    g = torch.Generator().manual_seed(2)
    x = torch.randn(Bn, D, L, generator=g, requires_grad=True)
    w = (torch.randn(D, W, generator=g) * 0.5).requires_grad_(True)
    b = torch.randn(D, generator=g).requires_grad_(True)
    ref = F.silu(F.conv1d(x, w.unsqueeze(1), b, padding=W - 1, groups=D)[..., :L])   # mamba_simple.py:318-321
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xc, wc, bc = (t.detach().cuda().requires_grad_(True) for t in (x, w, b))
    out = causal_conv1d_fn(xc, wc, bc, "silu")
    out.backward(dy.cuda())
    close(out, ref.detach(), 1e-5, "y")
    close(xc.grad, x.grad, 1e-5, "dx")
    close(wc.grad, w.grad, 2e-5, "dw")
    close(bc.grad, b.grad, 2e-5, "db")
    with pytest.raises(NotImplementedError):
        causal_conv1d_fn(xc, wc, bc, None)


def test_silu_gate_and_scan_with_z(hip_lib):
    from oracle.selective_scan import selective_scan_torch
    from nnuzoo_amd.selective_scan import selective_scan_fn
    g = torch.Generator().manual_seed(4)
    Bn, D, L, N = 2, 8, 70, 16
    u = torch.randn(Bn, D, L, generator=g)
    delta = torch.rand(Bn, D, L, generator=g) * 0.5
    A = -torch.rand(D, N, generator=g) - 0.1
    Bm, Cm = torch.randn(Bn, N, L, generator=g), torch.randn(Bn, N, L, generator=g)
    Dp, z, bias = torch.randn(D, generator=g), torch.randn(Bn, D, L, generator=g), torch.randn(D, generator=g) * 0.1
    leaves = [t.clone().requires_grad_(True) for t in (u, delta, A, Bm, Cm, Dp, z, bias)]
    # oracle scan (pinned to the reference's selective_scan_ref goldens) + the reference's gate `out * F.silu(z)`
    # (selective_scan_interface.py:146-147); B / C with one group: (b, 1, N, L)
    ref = selective_scan_torch(leaves[0], leaves[1], leaves[2], leaves[3][:, None], leaves[4][:, None], leaves[5],
                               delta_bias=leaves[7], delta_softplus=True) * F.silu(leaves[6])
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    dev = [t.detach().cuda().requires_grad_(True) for t in (u, delta, A, Bm, Cm, Dp, z, bias)]
    out = selective_scan_fn(*dev[:6], z=dev[6], delta_bias=dev[7], delta_softplus=True)
    out.backward(dy.cuda())
    close(out, ref.detach(), 3e-4, "y")
    for name, a, b in zip("u delta A B C D z bias".split(), dev, leaves):
        close(a.grad, b.grad, 1e-3, name)
