"""GPU parity of the tap-table conv kernels, norm, stem and head kernels (through the C-ABI) against torch CPU
fp32 on the same fp16-rounded inputs.  Tolerances: outputs are fp16 (rel 2^-11 rounding) of fp32 accumulations
whose summation order differs from the CPU's -> rtol 4e-3, atol scaled to the output magnitude."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from nnuzoo_amd import conv_plan as cp
from nnuzoo_amd import hip_ops as ops
from nnuzoo_amd.hip_ops import PreparedTable

DEV = "cuda"


def to_cl(x):  # (N, C, D, H, W) fp32 -> [N, V, C] fp16 on device
    N, C = x.shape[:2]
    return x.permute(0, 2, 3, 4, 1).reshape(N, -1, C).contiguous().to(torch.float16).to(DEV)


def from_cl(y, dims):  # [N, V, C] -> (N, C, D, H, W) fp32 cpu
    N, _, C = y.shape
    return y.float().cpu().reshape(N, *dims, C).permute(0, 4, 1, 2, 3).contiguous()


def h(x):  # fp16 rounding, kept in fp32
    return x.to(torch.float16).float()


def close(got, ref, rtol=4e-3, atol_frac=2e-3):
    atol = atol_frac * ref.abs().max().item() + 1e-6
    ok = torch.allclose(got, ref, rtol=rtol, atol=atol)
    if not ok:
        err = (got - ref).abs()
        idx = err.argmax()
        raise AssertionError(f"max err {err.max().item():.4g} (ref max {ref.abs().max().item():.4g}, atol {atol:.3g}) "
                             f"at flat {idx.item()}: got {got.flatten()[idx].item()} ref {ref.flatten()[idx].item()}; "
                             f"mismatch frac {(err > atol + rtol * ref.abs()).float().mean().item():.4f}")


CONV_CASES = [
    # N, dims, cin, cout, stride
    (2, (8, 16, 16), 32, 32, 1),
    (1, (12, 8, 24), 64, 32, 1),     # ragged vs the 4x8x8 / 8x8x8 tiles
    (1, (8, 8, 8), 32, 64, 1),
    (1, (4, 4, 4), 64, 128, 1),      # smaller than one tile
    (2, (16, 16, 16), 32, 64, 2),
    (1, (8, 8, 8), 64, 128, 2),
    (1, (64, 64, 64), 32, 32, 1),    # exercises the 8x8x8 tile
]


@pytest.mark.parametrize("N,dims,cin,cout,stride", CONV_CASES)
def test_conv_forward(hip_lib, N, dims, cin, cout, stride):
    g = torch.Generator().manual_seed(1)
    x = h(torch.randn(N, cin, *dims, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05)
    b = torch.randn(cout, generator=g)
    ref = F.conv3d(x, w, b, stride=stride, padding=1)
    odims = tuple(ref.shape[2:])
    pt = PreparedTable(cp.conv_forward(N, dims, cin, cout, stride=stride))
    wp = ops.pack_weight(w.to(DEV), pt, cin, cout, 27, cin * 27, 1)
    out = torch.full((N, int(np.prod(odims)), cout), float("nan"), dtype=torch.float16, device=DEV)
    ops.conv_tap_forward(pt, to_cl(x), wp, b.to(DEV), out)
    torch.cuda.synchronize()
    close(from_cl(out, odims), ref)
    # fused InstanceNorm statistics: {sum, sumsq} of exactly the fp16 values that were stored, added onto the buffer
    out2 = torch.empty_like(out)
    stats = torch.full((N, cout, 2), 3.0, dtype=torch.float32, device=DEV)
    ops.conv_tap_forward(pt, to_cl(x), wp, b.to(DEV), out2, stats=stats)
    torch.cuda.synchronize()
    assert torch.equal(out2, out)
    o64 = out.double().cpu()
    exp = torch.stack([o64.sum(1), (o64 * o64).sum(1)], dim=-1) + 3.0
    close(stats.cpu().double(), exp, rtol=1e-4, atol_frac=1e-5)


def test_conv_forward_strided_channels(hip_lib):
    """input read from / output written into channel slices of wider buffers (the zero-copy concat layout)."""
    g = torch.Generator().manual_seed(2)
    N, dims, cin, cout = 1, (8, 8, 16), 32, 32
    x = h(torch.randn(N, cin, *dims, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05)
    ref = F.conv3d(x, w, None, padding=1)
    V = int(np.prod(dims))
    xin = torch.randn(N, V, 2 * cin, generator=g).to(torch.float16).to(DEV)
    xin[:, :, cin:] = to_cl(x)
    out = torch.zeros((N, V, 3 * cout), dtype=torch.float16, device=DEV)
    pt = PreparedTable(cp.conv_forward(N, dims, cin, cout, ldi=2 * cin, ldo=3 * cout))
    wp = ops.pack_weight(w.to(DEV), pt, cin, cout, 27, cin * 27, 1)
    ops.conv_tap_forward(pt, xin[:, :, cin:], wp, None, out[:, :, cout:2 * cout])
    torch.cuda.synchronize()
    close(from_cl(out[:, :, cout:2 * cout].contiguous(), dims), ref)
    assert out[:, :, :cout].abs().max().item() == 0 and out[:, :, 2 * cout:].abs().max().item() == 0


@pytest.mark.parametrize("N,dims,cin,cout,stride", CONV_CASES[:6])
def test_conv_dgrad(hip_lib, N, dims, cin, cout, stride):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, cin, *dims, generator=g, requires_grad=True)
    w = h(torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05)
    y = F.conv3d(x, w, None, stride=stride, padding=1)
    dy = h(torch.randn(y.shape, generator=g))
    y.backward(dy)
    ref = x.grad
    pt = PreparedTable(cp.conv_dgrad(N, dims, cin, cout, stride=stride))
    wp = ops.pack_weight(w.to(DEV), pt, cout, cin, cin * 27, 27, 1)
    out = torch.full((N, int(np.prod(dims)), cin), float("nan"), dtype=torch.float16, device=DEV)
    ops.conv_tap_forward(pt, to_cl(dy), wp, None, out)
    torch.cuda.synchronize()
    close(from_cl(out, dims), ref)
    # accumulate variant: out += result
    base = h(torch.randn(N, cin, *dims, generator=g))
    out2 = to_cl(base)
    ops.conv_tap_forward(pt.with_accumulate(True), to_cl(dy), wp, None, out2)
    torch.cuda.synchronize()
    close(from_cl(out2, dims), ref + base)


@pytest.mark.parametrize("N,dims,cin,cout,stride", CONV_CASES[:6])
def test_conv_wgrad(hip_lib, N, dims, cin, cout, stride):
    g = torch.Generator().manual_seed(4)
    x = h(torch.randn(N, cin, *dims, generator=g))
    w = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    y = F.conv3d(x, w, None, stride=stride, padding=1)
    dy = h(torch.randn(y.shape, generator=g))
    y.backward(dy)
    ref = w.grad
    pt = PreparedTable(cp.conv_wgrad(N, dims, cin, cout, stride=stride))
    dw = torch.empty((27, cin, cout), dtype=torch.float32, device=DEV)
    ops.conv_tap_wgrad(pt, to_cl(x), to_cl(dy), dw)
    gw = torch.full((cout, cin, 3, 3, 3), float("nan"), dtype=torch.float32, device=DEV)
    ops.unpack_wgrad(dw, gw, cin, cout, 27, 27, cin * 27, 1, pt)
    torch.cuda.synchronize()
    close(gw.cpu(), ref, rtol=2e-3, atol_frac=1e-3)


@pytest.mark.parametrize("N,dims,cin,cout", [(2, (4, 4, 8), 64, 32), (1, (8, 8, 8), 128, 64), (1, (2, 2, 2), 320, 320)])
def test_conv_transpose(hip_lib, N, dims, cin, cout):
    g = torch.Generator().manual_seed(5)
    x = h(torch.randn(N, cin, *dims, generator=g)).requires_grad_(True)
    w = h(torch.randn(cin, cout, 2, 2, 2, generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(cout, generator=g)
    y = F.conv_transpose3d(x, w, b, stride=2)
    odims = tuple(y.shape[2:])
    dy = h(torch.randn(y.shape, generator=g))
    y.backward(dy)
    # forward, written into the first half of a 2*cout wide buffer
    V, Vo = int(np.prod(dims)), int(np.prod(odims))
    pt = PreparedTable(cp.convT_forward(N, dims, cin, cout, ldo=2 * cout))
    wd = w.detach().to(DEV)
    wp = ops.pack_weight(wd, pt, cin, cout, cout * 8, 8, 1)
    cat = torch.zeros((N, Vo, 2 * cout), dtype=torch.float16, device=DEV)
    ops.conv_tap_forward(pt, to_cl(x.detach()), wp, b.to(DEV), cat)
    torch.cuda.synchronize()
    close(from_cl(cat[:, :, :cout].contiguous(), odims), y.detach())
    assert cat[:, :, cout:].abs().max().item() == 0
    # dgrad
    gcat = torch.zeros((N, Vo, 2 * cout), dtype=torch.float16, device=DEV)
    gcat[:, :, :cout] = to_cl(dy)
    gcat[:, :, cout:] = 7.0  # must be ignored
    ptd = PreparedTable(cp.convT_dgrad(N, dims, cin, cout, ldi=2 * cout))
    wpd = ops.pack_weight(wd, ptd, cout, cin, 8, cout * 8, 1)
    dx = torch.full((N, V, cin), float("nan"), dtype=torch.float16, device=DEV)
    ops.conv_tap_forward(ptd, gcat, wpd, None, dx)
    torch.cuda.synchronize()
    close(from_cl(dx, dims), x.grad)
    # wgrad
    ptw = PreparedTable(cp.convT_wgrad(N, dims, cin, cout, lddout=2 * cout))
    dwt = torch.empty((8, cout, cin), dtype=torch.float32, device=DEV)
    ops.conv_tap_wgrad(ptw, gcat, to_cl(x.detach()), dwt)
    gw = torch.full((cin, cout, 2, 2, 2), float("nan"), dtype=torch.float32, device=DEV)
    ops.unpack_wgrad(dwt, gw, cout, cin, 8, 8, cout * 8, 1, ptw)
    torch.cuda.synchronize()
    close(gw.cpu(), w.grad, rtol=2e-3, atol_frac=1e-3)


@pytest.mark.parametrize("N,dims,C,ldy", [(2, (8, 8, 8), 32, 32), (1, (6, 10, 12), 64, 128), (2, (4, 4, 4), 320, 320)])
def test_instnorm_lrelu(hip_lib, N, dims, C, ldy):
    g = torch.Generator().manual_seed(6)
    x = (h(torch.randn(N, C, *dims, generator=g) * 2 + 0.5)).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.3).requires_grad_(True)
    y = F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), 0.01)
    dy = h(torch.randn(y.shape, generator=g))
    y.backward(dy)
    V = int(np.prod(dims))
    xr = to_cl(x.detach())
    stats = torch.empty((N, C, 2), dtype=torch.float32, device=DEV)
    ybuf = torch.zeros((N, V, ldy), dtype=torch.float16, device=DEV)
    yv = ybuf[:, :, ldy - C:]
    ops.instnorm_stats(xr, stats, N, V, C, C)
    ops.instnorm_lrelu_apply(xr, stats, gamma.detach().to(DEV), beta.detach().to(DEV), yv, N, V, C, C, ldy, 1e-5, 0.01)
    torch.cuda.synchronize()
    close(from_cl(yv.contiguous(), dims), y.detach(), rtol=3e-3, atol_frac=1e-3)
    red = torch.empty((N, C, 2), dtype=torch.float32, device=DEV)
    dx = torch.empty((N, V, C), dtype=torch.float16, device=DEV)
    gbuf = torch.zeros((N, V, ldy), dtype=torch.float16, device=DEV)
    gbuf[:, :, ldy - C:] = to_cl(dy)
    ops.instnorm_lrelu_bwd(xr, gbuf[:, :, ldy - C:], stats, gamma.detach().to(DEV), beta.detach().to(DEV), red, dx, N,
                           V, C, C, ldy, C, 1e-5, 0.01)
    torch.cuda.synchronize()
    close(from_cl(dx, dims), x.grad, rtol=5e-3, atol_frac=2e-3)
    rs = red.sum(0).cpu()
    close(rs[:, 1], gamma.grad, rtol=2e-3, atol_frac=1e-3)
    close(rs[:, 0], beta.grad, rtol=2e-3, atol_frac=1e-3)
    # the apply kernel can fold the per-sample reductions into the affine's parameter gradients itself
    dg, db = torch.full((C,), 9.0, device=DEV), torch.full((C,), 9.0, device=DEV)
    red2 = torch.empty_like(red)
    dx2 = torch.empty_like(dx)
    ops.instnorm_lrelu_bwd(xr, gbuf[:, :, ldy - C:], stats, gamma.detach().to(DEV), beta.detach().to(DEV), red2, dx2, N,
                           V, C, C, ldy, C, 1e-5, 0.01, dgamma=dg, dbeta=db)
    torch.cuda.synchronize()
    assert torch.equal(dx2, dx)
    close(dg.cpu(), gamma.grad, rtol=2e-3, atol_frac=1e-3)
    close(db.cpu(), beta.grad, rtol=2e-3, atol_frac=1e-3)


@pytest.mark.parametrize("N,dims", [(2, (8, 8, 8)), (1, (6, 12, 20))])
def test_stem(hip_lib, N, dims):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, 1, *dims, generator=g)
    w = (torch.randn(32, 1, 3, 3, 3, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(32, generator=g)
    y = F.conv3d(h(x), h(w), b, padding=1)
    dy = h(torch.randn(y.shape, generator=g))
    (gw_ref,) = torch.autograd.grad(F.conv3d(h(x), w, None, padding=1), w, dy)
    V = int(np.prod(dims))
    out = torch.empty((N, V, 32), dtype=torch.float16, device=DEV)
    ops.stem_forward(x.to(DEV), w.detach().to(DEV), b.to(DEV), out, (N, *dims), 32)
    torch.cuda.synchronize()
    close(from_cl(out, dims), y.detach())
    gw = torch.empty((32, 1, 3, 3, 3), dtype=torch.float32, device=DEV)
    ops.stem_wgrad(x.to(DEV), to_cl(dy), gw, (N, *dims), 32)
    torch.cuda.synchronize()
    close(gw.cpu(), gw_ref, rtol=2e-3, atol_frac=1e-3)


@pytest.mark.parametrize("N,dims,C,K", [(2, (8, 8, 8), 32, 2), (1, (5, 6, 7), 64, 3), (2, (4, 4, 4), 320, 2)])
def test_seg_head(hip_lib, N, dims, C, K):
    g = torch.Generator().manual_seed(8)
    x = h(torch.randn(N, C, *dims, generator=g)).requires_grad_(True)
    w = (torch.randn(K, C, 1, 1, 1, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(K, generator=g).requires_grad_(True)
    y = F.conv3d(x, h(w.detach()).requires_grad_(False) + (w - w.detach()), b)
    dl = h(torch.randn(y.shape, generator=g))
    y.backward(dl)
    V = int(np.prod(dims))
    xr = to_cl(x.detach())
    wd, bd = w.detach().to(DEV).reshape(K, C).contiguous(), b.detach().to(DEV)
    logits = torch.empty((N, K, *dims), dtype=torch.float16, device=DEV)
    ops.head_forward(xr, wd, bd, logits, N, V, C, K, C)
    torch.cuda.synchronize()
    close(logits.float().cpu(), y.detach())
    dlg = dl.to(torch.float16).to(DEV).contiguous()
    dx = torch.empty((N, V, C), dtype=torch.float16, device=DEV)
    ops.head_dgrad(dlg, wd, dx, N, V, C, K, C, False)
    gw = torch.empty((K, C), dtype=torch.float32, device=DEV)
    gb = torch.empty((K,), dtype=torch.float32, device=DEV)
    ops.head_wgrad(xr, dlg, gw, gb, N, V, C, K, C)
    torch.cuda.synchronize()
    close(from_cl(dx, dims), x.grad)
    close(gw.cpu(), w.grad.reshape(K, C), rtol=2e-3, atol_frac=1e-3)
    close(gb.cpu(), b.grad, rtol=2e-3, atol_frac=1e-3)


# ---- per-axis geometry: 2-D plans (depth-1 volumes) and anisotropic 3-D plans -----------------------------------------
ANISO_CASES = [
    # N, dims, cin, cout, ks, stride
    (2, (1, 40, 24), 32, 32, (1, 3, 3), (1, 1, 1)),      # 2-D, flat 1x32x8 tile, ragged
    (1, (1, 64, 64), 32, 64, (1, 3, 3), (1, 2, 2)),      # 2-D strided
    (2, (1, 16, 16), 64, 128, (1, 3, 3), (1, 1, 1)),     # 2-D small map -> 1x8x8 tile
    (1, (1, 12, 20), 64, 96, (1, 3, 3), (1, 2, 2)),      # 2-D strided, Cout % 64 != 0
    (1, (6, 16, 24), 32, 64, (1, 3, 3), (1, 2, 2)),      # thick-slice 3-D stage
    (1, (8, 16, 16), 64, 64, (3, 3, 3), (2, 2, 1)),      # pooling stopped on the last axis
    (1, (5, 16, 16), 32, 32, (1, 3, 3), (1, 1, 1)),      # k(1,3,3) s1 in 3-D
    (1, (8, 8, 12), 64, 64, (3, 3, 1), (2, 1, 2)),       # k1 s2 on one axis (uncovered odd positions)
]


def _w(cout, cin, ks, g, scale=0.05):
    return h(torch.randn(cout, cin, *ks, generator=g) * scale)


@pytest.mark.parametrize("N,dims,cin,cout,ks,stride", ANISO_CASES)
def test_conv_aniso_forward_dgrad_wgrad(hip_lib, N, dims, cin, cout, ks, stride):
    g = torch.Generator().manual_seed(11)
    nk = ks[0] * ks[1] * ks[2]
    pad = [k // 2 for k in ks]
    x = h(torch.randn(N, cin, *dims, generator=g)).requires_grad_(True)
    w = _w(cout, cin, ks, g).requires_grad_(True)
    b = torch.randn(cout, generator=g)
    y = F.conv3d(x, w, b, stride=stride, padding=pad)
    odims = tuple(y.shape[2:])
    dy = h(torch.randn(y.shape, generator=g))
    y.backward(dy)
    wd = w.detach().to(DEV)
    # forward
    pt = PreparedTable(cp.conv_forward(N, dims, cin, cout, ks=ks, stride=stride))
    wp = ops.pack_weight(wd, pt, cin, cout, nk, cin * nk, 1)
    out = torch.full((N, int(np.prod(odims)), cout), float("nan"), dtype=torch.float16, device=DEV)
    ops.conv_tap_forward(pt, to_cl(x.detach()), wp, b.to(DEV), out)
    torch.cuda.synchronize()
    close(from_cl(out, odims), y.detach())
    # dgrad (destination zeroed when the table leaves positions unwritten)
    ptd = PreparedTable(cp.conv_dgrad(N, dims, cin, cout, ks=ks, stride=stride))
    wpd = ops.pack_weight(wd, ptd, cout, cin, cin * nk, nk, 1)
    fill = 0.0 if cp.dgrad_uncovered(ks, stride) else float("nan")
    dx = torch.full((N, int(np.prod(dims)), cin), fill, dtype=torch.float16, device=DEV)
    ops.conv_tap_forward(ptd, to_cl(dy), wpd, None, dx)
    torch.cuda.synchronize()
    close(from_cl(dx, dims), x.grad)
    # wgrad
    ptw = PreparedTable(cp.conv_wgrad(N, dims, cin, cout, ks=ks, stride=stride))
    dw = torch.empty((nk, cin, cout), dtype=torch.float32, device=DEV)
    ops.conv_tap_wgrad(ptw, to_cl(x.detach()), to_cl(dy), dw)
    gw = torch.full((cout, cin, *ks), float("nan"), dtype=torch.float32, device=DEV)
    ops.unpack_wgrad(dw, gw, cin, cout, nk, nk, cin * nk, 1, ptw)
    torch.cuda.synchronize()
    close(gw.cpu(), w.grad, rtol=2e-3, atol_frac=1e-3)


@pytest.mark.parametrize("N,dims,cin,cout,stride", [(2, (1, 12, 20), 64, 32, (1, 2, 2)), (1, (4, 6, 8), 128, 64, (1, 2, 2)),
                                                    (1, (4, 4, 8), 64, 64, (2, 2, 1))])
def test_conv_transpose_aniso(hip_lib, N, dims, cin, cout, stride):
    g = torch.Generator().manual_seed(12)
    nk = stride[0] * stride[1] * stride[2]
    x = h(torch.randn(N, cin, *dims, generator=g)).requires_grad_(True)
    w = h(torch.randn(cin, cout, *stride, generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(cout, generator=g)
    y = F.conv_transpose3d(x, w, b, stride=stride)
    odims = tuple(y.shape[2:])
    dy = h(torch.randn(y.shape, generator=g))
    y.backward(dy)
    V, Vo = int(np.prod(dims)), int(np.prod(odims))
    wd = w.detach().to(DEV)
    pt = PreparedTable(cp.convT_forward(N, dims, cin, cout, stride=stride))
    wp = ops.pack_weight(wd, pt, cin, cout, cout * nk, nk, 1)
    out = torch.full((N, Vo, cout), float("nan"), dtype=torch.float16, device=DEV)
    ops.conv_tap_forward(pt, to_cl(x.detach()), wp, b.to(DEV), out)
    torch.cuda.synchronize()
    close(from_cl(out, odims), y.detach())
    ptd = PreparedTable(cp.convT_dgrad(N, dims, cin, cout, stride=stride))
    wpd = ops.pack_weight(wd, ptd, cout, cin, nk, cout * nk, 1)
    dx = torch.full((N, V, cin), float("nan"), dtype=torch.float16, device=DEV)
    ops.conv_tap_forward(ptd, to_cl(dy), wpd, None, dx)
    torch.cuda.synchronize()
    close(from_cl(dx, dims), x.grad)
    ptw = PreparedTable(cp.convT_wgrad(N, dims, cin, cout, stride=stride))
    dwt = torch.empty((nk, cout, cin), dtype=torch.float32, device=DEV)
    ops.conv_tap_wgrad(ptw, to_cl(dy), to_cl(x.detach()), dwt)
    gw = torch.full((cin, cout, *stride), float("nan"), dtype=torch.float32, device=DEV)
    ops.unpack_wgrad(dwt, gw, cout, cin, nk, nk, cout * nk, 1, ptw)
    torch.cuda.synchronize()
    close(gw.cpu(), w.grad, rtol=2e-3, atol_frac=1e-3)


def test_batched_weight_pack_matches_single(hip_lib):
    """one launch packing several layers (all four parameter layouts of the schedule) == per-layer packing, bit-exact"""
    g = torch.Generator().manual_seed(13)
    jobs = ops.PackJobTable(torch.device(DEV))
    expect = []
    for cin, cout, ks in [(32, 64, (3, 3, 3)), (96, 32, (1, 3, 3)), (64, 320, (3, 3, 3))]:
        nk = ks[0] * ks[1] * ks[2]
        w = torch.randn(cout, cin, *ks, generator=g).to(DEV)
        for pt, R, Cc, sr, sc in [(PreparedTable(cp.conv_forward(1, (4, 8, 8), cin, cout, ks=ks)), cin, cout, nk, cin * nk),
                                  (PreparedTable(cp.conv_dgrad(1, (4, 8, 8), cin, cout, ks=ks)), cout, cin, cin * nk, nk)]:
            dst = torch.zeros(cin * cout * nk, dtype=torch.float16, device=DEV)
            jobs.add(w, dst, pt, R, Cc, sr, sc, 1)
            expect.append((dst, ops.pack_weight(w, pt, R, Cc, sr, sc, 1)))
    wt = torch.randn(128, 64, 2, 2, 2, generator=g).to(DEV)
    for pt, R, Cc, sr, sc in [(PreparedTable(cp.convT_forward(1, (4, 4, 4), 128, 64)), 128, 64, 64 * 8, 8),
                              (PreparedTable(cp.convT_dgrad(1, (4, 4, 4), 128, 64)), 64, 128, 8, 64 * 8)]:
        dst = torch.zeros(128 * 64 * 8, dtype=torch.float16, device=DEV)
        jobs.add(wt, dst, pt, R, Cc, sr, sc, 1)
        expect.append((dst, ops.pack_weight(wt, pt, R, Cc, sr, sc, 1)))
    jobs.run()
    torch.cuda.synchronize()
    for got, ref in expect:
        assert torch.equal(got, ref)


@pytest.mark.parametrize("N,dims,cin,cout,stride", [(2, (4, 4, 4), 320, 320, 1), (2, (8, 8, 8), 640, 320, 1),
                                                    (1, (8, 8, 8), 256, 320, 2), (2, (6, 5, 7), 128, 64, 1),
                                                    (2, (16, 16, 16), 256, 256, 1)])
def test_conv_splitk_small_levels(hip_lib, N, dims, cin, cout, stride):
    """round 3: the <= 8^3 levels split the reduction over workgroups when a workspace is handed in (split-K over
    16-channel slices + a finishing kernel): forward with the fused InstanceNorm table, data gradient plain and
    accumulating - against torch fp32 on the same fp16-rounded operands and against the unsplit launch; run twice:
    bit-identical (fixed split order)"""
    from nnuzoo_amd._lib import call, load
    g = torch.Generator().manual_seed(cin + dims[0])
    x = h(torch.randn(N, cin, *dims, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5)
    b = torch.randn(cout, generator=g)
    ref = F.conv3d(x, w, b, stride=stride, padding=1)
    odims = tuple(ref.shape[2:])
    V = int(np.prod(odims))
    pt = PreparedTable(cp.conv_forward(N, dims, cin, cout, stride=stride))
    wp = ops.pack_weight(w.to(DEV), pt, cin, cout, 27, cin * 27, 1)
    ws = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
    xc = to_cl(x)
    plain = torch.empty((N, V, cout), dtype=torch.float16, device=DEV)
    ops.conv_tap_forward(pt, xc, wp, b.to(DEV), plain)
    gamma, beta = (1 + 0.1 * torch.randn(cout, generator=g)).to(DEV), (0.1 * torch.randn(cout, generator=g)).to(DEV)
    sc = ops.NormScratch(torch.device(DEV), N * cout)
    outs = []
    for _ in range(2):
        out = torch.full((N, V, cout), float("nan"), dtype=torch.float16, device=DEV)
        nstat = torch.full((N, cout, 4), float("nan"), device=DEV)
        ops.conv_tap_forward_norm(pt, xc, wp, b.to(DEV), out, sc, gamma, beta, 1e-5, nstat, workspace=ws)
        outs.append((out, nstat))
    torch.cuda.synchronize()
    close(from_cl(outs[0][0], odims), ref)
    assert (outs[0][0].float() - plain.float()).abs().max().item() <= 2e-3 * plain.float().abs().max().item()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    o64 = outs[0][0].double()
    mean, var = o64.mean(1), o64.var(1, unbiased=False)
    rstd = 1 / torch.sqrt(var + 1e-5)
    tab = outs[0][1].double()
    # (rstd / scale: where the table comes from the conv epilogue's matrix-core moments - nnz_conv_tuning knob 11, the default - the
    #  variance is that of fp16-rounded deviations: within 1e-4 relative; the split-K finishing kernel and the VALU form hold 1e-5.
    #  The mean is exact in all three.)
    assert torch.allclose(tab[..., 0], mean, rtol=1e-5, atol=1e-6) and torch.allclose(tab[..., 1], rstd, rtol=1e-4)
    assert torch.allclose(tab[..., 2], rstd * gamma.double(), rtol=1e-4)
    assert torch.allclose(tab[..., 3], beta.double() - mean * rstd * gamma.double(), rtol=1e-4, atol=1e-5)
    # data gradient (one tap group at stride 1): plain and accumulating
    if stride == 1:
        dy = h(torch.randn(N, cout, *odims, generator=g))
        xr = x.clone().requires_grad_(True)
        (rdx,) = torch.autograd.grad(F.conv3d(xr, w, None, stride=1, padding=1), xr, dy)
        ptd = PreparedTable(cp.conv_dgrad(N, dims, cin, cout, stride=1))
        wpd = ops.pack_weight(w.to(DEV), ptd, cout, cin, cin * 27, 27, 1)
        dx = torch.full((N, int(np.prod(dims)), cin), float("nan"), dtype=torch.float16, device=DEV)
        ops.conv_tap_forward(ptd, to_cl(dy), wpd, None, dx, workspace=ws)
        close(from_cl(dx, dims), rdx)
        base = h(torch.randn(N, cin, *dims, generator=g))
        acc = to_cl(base).clone()
        ops.conv_tap_forward(ptd.with_accumulate(True), to_cl(dy), wpd, None, acc, workspace=ws)
        close(from_cl(acc, dims), rdx + base)
    lib = load()
    call("nnz_conv_tuning", 5, 1)                   # the A/B switch: split-K off -> the unsplit kernel, same interface
    try:
        out = torch.empty_like(plain)
        ops.conv_tap_forward(pt, xc, wp, b.to(DEV), out, workspace=ws)
        assert torch.equal(out, plain)
    finally:
        call("nnz_conv_tuning", 5, 0)


def test_dual_weight_pack_matches_single(hip_lib):
    """round 3: ONE launch reading every parameter once and writing the forward and the data-gradient packed forms
    (csrc/conv_pack.hip pack_dual_kernel) == the two per-layer packs, bit-exact; conv (3-D k3, stride 2, thick-slice k(1,3,3),
    2-D) and transposed-conv parameter layouts, several 32 x 32 blocks per layer"""
    g = torch.Generator().manual_seed(17)
    dual = ops.DualPackTable(torch.device(DEV))
    expect = []
    for cin, cout, ks, st in [(32, 64, (3, 3, 3), 1), (96, 32, (1, 3, 3), 1), (64, 320, (3, 3, 3), 2), (32, 32, (3, 3, 3), 1),
                              (640, 320, (3, 3, 3), 1)]:
        nk = ks[0] * ks[1] * ks[2]
        w = torch.randn(cout, cin, *ks, generator=g).to(DEV)
        st3 = (1, st, st) if ks[0] == 1 else (st,) * 3
        fwd = PreparedTable(cp.conv_forward(1, (4, 8, 8), cin, cout, ks=ks, stride=st3))
        dgr = PreparedTable(cp.conv_dgrad(1, (4, 8, 8), cin, cout, ks=ks, stride=st3))
        df = torch.zeros(cin * cout * nk, dtype=torch.float16, device=DEV)
        db = torch.zeros(cin * cout * nk, dtype=torch.float16, device=DEV)
        dual.add(w, df, db, cin, cout, nk, True, fwd, dgr)
        expect.append((df, ops.pack_weight(w, fwd, cin, cout, nk, cin * nk, 1)))
        expect.append((db, ops.pack_weight(w, dgr, cout, cin, cin * nk, nk, 1)))
    for cin, cout, st in [(128, 64, (2, 2, 2)), (64, 32, (1, 2, 2))]:
        nk = st[0] * st[1] * st[2]
        wt = torch.randn(cin, cout, *st, generator=g).to(DEV)
        fwd = PreparedTable(cp.convT_forward(1, (4, 4, 4), cin, cout, stride=st))
        dgr = PreparedTable(cp.convT_dgrad(1, (4, 4, 4), cin, cout, stride=st))
        df = torch.zeros(cin * cout * nk, dtype=torch.float16, device=DEV)
        db = torch.zeros(cin * cout * nk, dtype=torch.float16, device=DEV)
        dual.add(wt, df, db, cin, cout, nk, False, fwd, dgr)
        expect.append((df, ops.pack_weight(wt, fwd, cin, cout, cout * nk, nk, 1)))
        expect.append((db, ops.pack_weight(wt, dgr, cout, cin, nk, cout * nk, 1)))
    dual.run()
    torch.cuda.synchronize()
    for i, (got, ref) in enumerate(expect):
        assert torch.equal(got, ref), i


def test_conv_batches_beyond_2g_elements(hip_lib):
    """tensors with more than 2^31 elements (5 samples x 128^3 voxels x channel stride 256): the entry points split the
    batch into sample chunks (32-bit voxel offsets inside the kernels); forward + statistics and weight gradient."""
    g = torch.Generator().manual_seed(21)
    N, dims, C, ld = 5, (128, 128, 128), 32, 256
    V = int(np.prod(dims))
    x1 = h(torch.randn(1, C, *dims, generator=g))
    w = h(torch.randn(C, C, 3, 3, 3, generator=g) * 0.05)
    ref = F.conv3d(x1, w, None, padding=1)
    xin = torch.zeros((N, V, ld), dtype=torch.float16, device=DEV)
    xin[:, :, 64:64 + C] = to_cl(x1)                      # every sample holds the same patch in channels 64..95
    out = torch.zeros((N, V, ld), dtype=torch.float16, device=DEV)
    pt = PreparedTable(cp.conv_forward(N, dims, C, C, ldi=ld, ldo=ld))
    wp = ops.pack_weight(w.to(DEV), pt, C, C, 27, C * 27, 1)
    stats = torch.zeros((N, C, 2), dtype=torch.float32, device=DEV)
    ops.conv_tap_forward(pt, xin[:, :, 64:], wp, None, out[:, :, 128:], stats=stats)
    torch.cuda.synchronize()
    got0 = from_cl(out[0:1, :, 128:128 + C].contiguous(), dims)
    close(got0, ref)
    for n in range(1, N):
        assert torch.equal(out[n], out[0]), n
    assert out[:, :, :128].abs().max().item() == 0 and out[:, :, 128 + C:].abs().max().item() == 0
    # (fp32 atomics: the per-sample sums agree to summation-order noise, absolute in the scale of the sum of squares)
    assert torch.allclose(stats[1:], stats[:1].expand(N - 1, C, 2), rtol=1e-5, atol=1e-6 * stats.abs().max().item())
    # weight gradient over the 5 chunks' samples = 5 x the single-sample gradient
    dy1 = h(torch.randn(ref.shape, generator=g))
    wz = torch.zeros(C, C, 3, 3, 3, requires_grad=True)
    F.conv3d(x1, wz, None, padding=1).backward(dy1)
    dyb = torch.zeros((N, V, ld), dtype=torch.float16, device=DEV)
    dyb[:, :, :C] = to_cl(dy1)
    ptw = PreparedTable(cp.conv_wgrad(N, dims, C, C, ldx=ld, lddy=ld))
    dw = torch.empty((27, C, C), dtype=torch.float32, device=DEV)
    ops.conv_tap_wgrad(ptw, xin[:, :, 64:], dyb, dw)
    gw = torch.empty((C, C, 3, 3, 3), dtype=torch.float32, device=DEV)
    ops.unpack_wgrad(dw, gw, C, C, 27, 27, C * 27, 1, ptw)
    torch.cuda.synchronize()
    close(gw.cpu(), N * wz.grad, rtol=2e-3, atol_frac=1e-3)


@pytest.mark.parametrize("N,dims,cin,cout,stride", [CONV_CASES[0], CONV_CASES[2], CONV_CASES[4], (2, (32, 32, 32), 32, 32, 1)])
def test_conv_wgrad_two_stage_deterministic(hip_lib, N, dims, cin, cout, stride):
    """partial blocks + fixed-order reduction straight into the torch-layout gradient: equals the reference gradient,
    is bit-identical run to run, and honours `accumulate`"""
    g = torch.Generator().manual_seed(31)
    x = h(torch.randn(N, cin, *dims, generator=g))
    w = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    y = F.conv3d(x, w, None, stride=stride, padding=1)
    dy = h(torch.randn(y.shape, generator=g))
    y.backward(dy)
    pt = PreparedTable(cp.conv_wgrad(N, dims, cin, cout, stride=stride))
    ws = torch.empty(ops.conv_tap_wgrad_workspace_floats(pt), dtype=torch.float32, device=DEV)
    xs, dys = to_cl(x), to_cl(dy)
    outs = []
    for rep in range(3):
        ws.fill_(float("nan"))                      # nothing may depend on the workspace's previous content
        gw = torch.full((cout, cin, 3, 3, 3), float("nan"), dtype=torch.float32, device=DEV)
        ops.conv_tap_wgrad_to_grad(pt, xs, dys, ws, gw, 27, cin * 27, 1)
        torch.cuda.synchronize()
        outs.append(gw)
    close(outs[0].cpu(), w.grad, rtol=2e-3, atol_frac=1e-3)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    acc = torch.ones_like(outs[0])
    ops.conv_tap_wgrad_to_grad(pt, xs, dys, ws, acc, 27, cin * 27, 1, accumulate=True)
    torch.cuda.synchronize()
    assert torch.allclose(acc, outs[0] + 1.0, rtol=0, atol=1e-6 * outs[0].abs().max().item() + 1e-6)


@pytest.mark.parametrize("N,dims,cin,cout,stride,ld_out_mult", [
    (2, (8, 8, 8), 64, 32, (2, 2, 2), 2),        # the full-resolution stage's shape family; output = channel slice of a cat buffer
    (1, (5, 6, 7), 128, 64, (2, 2, 2), 2),       # voxel count not a multiple of the 32-voxel tile
    (2, (4, 9, 8), 64, 32, (1, 2, 2), 1),        # anisotropic stride
    (3, (1, 12, 10), 64, 32, (1, 2, 2), 2),      # 2-D plan (depth-1 volume)
    (1, (3, 4, 5), 32, 96, (2, 2, 2), 1)])
def test_conv_transpose_kernel_matches_torch(hip_lib, N, dims, cin, cout, stride, ld_out_mult):
    """csrc/conv_transpose.hip (kernel = stride ConvTranspose, forward + data gradient) against torch's ConvTranspose3d in
    fp32 on the fp16-rounded operands"""
    from nnuzoo_amd import hip_ops as ops
    torch.manual_seed(0)
    assert ops.convT_supported(cin, cout, stride, False)
    m = torch.nn.ConvTranspose3d(cin, cout, stride, stride, bias=True).cuda()
    V = dims[0] * dims[1] * dims[2]
    od = tuple(d * s for d, s in zip(dims, stride))
    Vo = od[0] * od[1] * od[2]
    x = torch.randn(N, V, cin, device="cuda").half()
    ldo = cout * ld_out_mult
    out = torch.full((N, Vo, ldo), 7.0, device="cuda", dtype=torch.float16)
    ops.convT_forward(x, m.weight.detach(), m.bias.detach(), out, N, dims, cin, cout, stride, cin, ldo)
    w16 = m.weight.detach().half().float()
    xr = x.float().transpose(1, 2).reshape(N, cin, *dims)
    ref = torch.nn.functional.conv_transpose3d(xr, w16, m.bias.detach(), stride=stride)
    ref_cl = ref.reshape(N, cout, Vo).transpose(1, 2)
    got = out[..., :cout].float()
    assert torch.allclose(got, ref_cl, rtol=2e-3, atol=2e-3 * ref_cl.abs().max().item())
    if ld_out_mult > 1:
        assert bool((out[..., cout:] == 7.0).all())          # the other half of the wider buffer is untouched
    if ops.convT_supported(cin, cout, stride, True):
        dy = torch.randn(N, Vo, ldo, device="cuda").half()
        din = torch.empty(N, V, cin, device="cuda", dtype=torch.float16)
        ops.convT_dgrad(dy, m.weight.detach(), din, N, dims, cin, cout, stride, ldo, cin)
        dyr = dy[..., :cout].float().transpose(1, 2).reshape(N, cout, *od)
        ref_d = torch.nn.functional.conv3d(dyr, w16, stride=stride)          # adjoint of the transposed conv
        ref_d = ref_d.reshape(N, cin, V).transpose(1, 2)
        assert torch.allclose(din.float(), ref_d, rtol=2e-3, atol=2e-3 * ref_d.abs().max().item())
