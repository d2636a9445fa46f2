"""CPU tests of the host-side tap tables (nnuzoo_amd/conv_plan.py): a slow, literal interpreter of the
nnz_conv_desc semantics (include/nnuzoo_hip.h) is run on the tables and compared with torch's conv ops.
Exactly what the HIP kernels are specified to compute, so a failure here is a host bug, not a kernel bug."""
import itertools

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from nnuzoo_amd import conv_plan as cp


def _gather(x, m_dims, stride, off):
    """x: (N, D, H, W, C) -> values at m*stride + off for all m (zero outside)."""
    N, D, H, W, C = x.shape
    out = torch.zeros((N, *m_dims, C), dtype=x.dtype)
    idx = []
    for a, (n_m, dim) in enumerate(zip(m_dims, (D, H, W))):
        coords = torch.arange(n_m) * stride[a] + off[a]
        ok = (coords >= 0) & (coords < dim)
        idx.append((coords, ok))
    sel = [torch.nonzero(ok).flatten() for _, ok in idx]
    if any(len(s) == 0 for s in sel):
        return out
    src = [idx[a][0][sel[a]] for a in range(3)]
    out[:, sel[0][:, None, None], sel[1][None, :, None], sel[2][None, None, :]] = \
        x[:, src[0][:, None, None], src[1][None, :, None], src[2][None, None, :]]
    return out


def interp_forward(t: cp.TapTable, x, wslices, bias=None, out=None):
    """x (N, *in_dims, Cin); wslices[j] (Cin, Cout) for tap position j in table order."""
    if out is None:
        out = torch.zeros((t.N, *t.out_dims, t.Cout), dtype=torch.float64)
    j = 0
    for ooff, taps in t.groups:
        acc = torch.zeros((t.N, *t.m_dims, t.Cout), dtype=torch.float64)
        if bias is not None:
            acc += bias
        for off, _ in taps:
            acc += _gather(x, t.m_dims, t.in_stride, off) @ wslices[j]
            j += 1
        for m in itertools.product(*[range(d) for d in t.m_dims]):
            o = tuple(m[a] * t.out_stride[a] + ooff[a] for a in range(3))
            if all(o[a] < t.out_dims[a] for a in range(3)):
                if t.accumulate:
                    out[:, o[0], o[1], o[2]] += acc[:, m[0], m[1], m[2]]
                else:
                    out[:, o[0], o[1], o[2]] = acc[:, m[0], m[1], m[2]]
    return out


def interp_wgrad(t: cp.TapTable, boxed, plain):
    """dW[widx][a][b] = sum_m boxed[m*IS + off][a] * plain[m*OS + ooff][b]."""
    assert len(t.groups) == 1
    ooff, taps = t.groups[0]
    T = max(w for _, w in taps) + 1
    dw = torch.zeros((T, t.Cin, t.Cout), dtype=torch.float64)
    q = _gather(plain, t.m_dims, t.out_stride, ooff)
    for off, widx in taps:
        p = _gather(boxed, t.m_dims, t.in_stride, off)
        dw[widx] += torch.einsum("ndhwa,ndhwb->ab", p, q)
    return dw


def cl(x):
    return x.permute(0, 2, 3, 4, 1).contiguous().double()


def ncdhw(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize("dims,ks,stride", [
    ((4, 6, 8), 3, 1), ((4, 6, 8), 3, 2), ((5, 7, 6), 3, 2), ((2, 2, 2), 3, 1),
    ((4, 6, 8), (1, 3, 3), (1, 2, 2)),      # anisotropic 3-D stage (nnU-Net planner, thick slices)
    ((4, 6, 8), (3, 3, 3), (2, 2, 1)),      # pooling stopped on the last axis
    ((4, 6, 7), (3, 3, 1), (2, 1, 2)),      # k1 s2 on one axis: odd input positions receive no gradient
    ((1, 10, 12), (1, 3, 3), (1, 1, 1)),    # 2-D layer = depth-1 volume
    ((1, 9, 12), (1, 3, 3), (1, 2, 2)),
])
def test_conv_tables(dims, ks, stride):
    g = torch.Generator().manual_seed(0)
    N, cin, cout = 2, 3, 4
    ks3, st3 = cp._triple(ks), cp._triple(stride)
    nk = ks3[0] * ks3[1] * ks3[2]
    x = torch.randn(N, cin, *dims, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cout, cin, *ks3, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(cout, generator=g, dtype=torch.float64)
    y = F.conv3d(x, w, b, stride=st3, padding=[k // 2 for k in ks3])
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    wflat = w.detach().reshape(cout, cin, nk)
    # forward
    t = cp.conv_forward(N, dims, cin, cout, ks=ks, stride=stride)
    assert t.out_dims == tuple(y.shape[2:])
    ws = [wflat[:, :, k].t() for k in t.pack_ksel]
    got = interp_forward(t, cl(x.detach()), ws, b)
    assert torch.allclose(ncdhw(got), y.detach(), atol=1e-10)
    # dgrad (destination zeroed first: interp_forward starts from zeros, as the caller must when dgrad_uncovered)
    t = cp.conv_dgrad(N, dims, cin, cout, ks=ks, stride=stride)
    ws = [wflat[:, :, k] for k in t.pack_ksel]  # (Cout, Cin): reduction over cout
    got = interp_forward(t, cl(dy), ws)
    assert torch.allclose(ncdhw(got), x.grad, atol=1e-10)
    d = t.to_desc()
    assert d.ntaps_total == nk
    assert list(d.ext) == [(ks3[a] - 1) if st3[a] == 1 else (1 if ks3[a] == 3 else 0) for a in range(3)]
    assert cp.dgrad_uncovered(ks, stride) == any(st3[a] == 2 and ks3[a] == 1 for a in range(3))
    # wgrad
    t = cp.conv_wgrad(N, dims, cin, cout, ks=ks, stride=stride)
    dw = interp_wgrad(t, cl(x.detach()), cl(dy))  # [nk][cin][cout]
    assert torch.allclose(dw.permute(2, 1, 0).reshape(cout, cin, *ks3), w.grad, atol=1e-10)
    if st3 == (1, 1, 1):
        # round 4: the same gradient with the operand roles exchanged (boxed = dY, plain = X; block [nk][cout][cin])
        t = cp.conv_wgrad_flipped(N, dims, cin, cout, ks=ks)
        assert (t.Cin, t.Cout) == (cout, cin)
        dwf = interp_wgrad(t, cl(dy), cl(x.detach()))
        assert torch.allclose(dwf.permute(1, 2, 0).reshape(cout, cin, *ks3), w.grad, atol=1e-10)


@pytest.mark.parametrize("dims,stride", [((2, 3, 4), 2), ((1, 1, 1), 2), ((2, 3, 4), (1, 2, 2)), ((1, 5, 4), (1, 2, 2)),
                                         ((3, 2, 4), (2, 2, 1))])
def test_conv_transpose_tables(dims, stride):
    g = torch.Generator().manual_seed(1)
    N, cin, cout = 2, 3, 5
    st = cp._triple(stride)
    nk = st[0] * st[1] * st[2]
    x = torch.randn(N, cin, *dims, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cin, cout, *st, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(cout, generator=g, dtype=torch.float64)
    y = F.conv_transpose3d(x, w, b, stride=st)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    wflat = w.detach().reshape(cin, cout, nk)
    t = cp.convT_forward(N, dims, cin, cout, stride=stride)
    got = interp_forward(t, cl(x.detach()), [wflat[:, :, k] for k in t.pack_ksel], b)
    assert torch.allclose(ncdhw(got), y.detach(), atol=1e-10)
    t = cp.convT_dgrad(N, dims, cin, cout, stride=stride)
    got = interp_forward(t, cl(dy), [wflat[:, :, k].t() for k in t.pack_ksel])
    assert torch.allclose(ncdhw(got), x.grad, atol=1e-10)
    t = cp.convT_wgrad(N, dims, cin, cout, stride=stride)
    dw = interp_wgrad(t, cl(dy), cl(x.detach()))  # [nk][cout][cin]
    assert torch.allclose(dw.permute(2, 1, 0).reshape(cin, cout, *st), w.grad, atol=1e-10)


def test_desc_roundtrip_limits():
    t = cp.conv_dgrad(1, (8, 8, 8), 32, 64, stride=2)
    d = t.to_desc()
    assert d.ngroups == 8 and list(d.out_stride) == [2, 2, 2] and list(d.in_stride) == [1, 1, 1]
    assert sorted(g.ntaps for g in list(d.groups)[:8]) == [1, 2, 2, 2, 4, 4, 4, 8]
    assert [d.lo[i] for i in range(3)] == [0, 0, 0]
