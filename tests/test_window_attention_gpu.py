"""Swin window-attention core (csrc/window_attention.hip, round-3 kernels: four waves per workgroup, operands straight
from global memory, bias matrix per workgroup, deterministic bias-table gradient) against the explicit float64 formula
of the reference's WindowAttention.forward (/root/reference/nnunetv2/nets/swt2net.py:584-619: roll, 7x7 partition,
softmax(q k^T * scale + table[index] (+ -100 mask)) v, merge, roll back) on the shapes of the four SwT2Net stages
(head_dim 32, heads 3 / 6 / 12 / 24 at 512^2 -> H, W = 133 / 70 / 35 / 21 after padding; here smaller grids with the
same head_dim and the window-run logic: windows per workgroup 4 ... 16, ragged last run), odd head_dims (2, 8, 12, 16, 24)
and both shift settings.  The reference's own goldens are in test_zoo_gpu.py.  Also: the bias-table gradient and dqkv are
bit-identical across two runs (no float atomics)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from nnuzoo_amd.window_attention import window_attention_core

DEV = "cuda"


def _index():
    ar = torch.arange(7)
    yy, xx = torch.meshgrid(ar, ar, indexing="ij")
    y, x = yy.flatten(), xx.flatten()
    return ((y[:, None] - y[None, :] + 6) * 13 + (x[:, None] - x[None, :] + 6)).to(torch.int32)


def _reference(qkv, table, index, heads, shift, scale):
    """float64, the reference's op sequence"""
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    hd = C // heads
    x = qkv
    if shift:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    nh, nw = H // 7, W // 7
    xw = x.view(B, nh, 7, nw, 7, 3, heads, hd).permute(5, 0, 1, 3, 6, 2, 4, 7).reshape(3, B * nh * nw, heads, 49, hd)
    q, k, v = xw[0] * scale, xw[1], xw[2]
    att = q @ k.transpose(-1, -2) + table[index.long().reshape(-1)].view(49, 49, heads).permute(2, 0, 1)[None]
    if shift:
        img = torch.zeros(H, W)
        cnt = 0
        for hs in (slice(0, -7), slice(-7, -shift), slice(-shift, None)):
            for ws in (slice(0, -7), slice(-7, -shift), slice(-shift, None)):
                img[hs, ws] = cnt
                cnt += 1
        mw = img.view(nh, 7, nw, 7).permute(0, 2, 1, 3).reshape(nh * nw, 49)
        mask = (mw[:, None, :] - mw[:, :, None] != 0).to(qkv.dtype) * -100.0          # (nW, 49, 49)
        att = att.view(B, nh * nw, heads, 49, 49) + mask[None, :, None].to(qkv.device)
        att = att.view(B * nh * nw, heads, 49, 49)
    o = torch.softmax(att, -1) @ v                                                    # (Bn, heads, 49, hd)
    o = o.view(B, nh, nw, heads, 7, 7, hd).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, H, W, C)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o


@pytest.mark.parametrize("B,H,W,heads,hd,shift", [(2, 21, 21, 3, 32, 0), (2, 21, 21, 3, 32, 3), (1, 35, 28, 6, 32, 3),
                                                   (2, 14, 14, 24, 32, 3), (1, 14, 14, 2, 2, 3), (1, 7, 7, 1, 8, 0),
                                                   (3, 14, 21, 4, 12, 3), (1, 70, 70, 2, 16, 3), (2, 28, 28, 5, 24, 0),
                                                   (2, 133, 133, 3, 32, 3)])
def test_against_float64_formula(hip_lib, B, H, W, heads, hd, shift):
    g = torch.Generator().manual_seed(H * 31 + heads)
    C = heads * hd
    qkv = torch.randn(B, H, W, 3 * C, generator=g)
    table = torch.randn(169, heads, generator=g) * 0.5
    dout = torch.randn(B, H, W, C, generator=g)
    idx = _index()
    scale = hd ** -0.5
    q64 = qkv.double().requires_grad_(True)
    t64 = table.double().requires_grad_(True)
    ref = _reference(q64, t64, idx, heads, shift, scale)
    rq, rt = torch.autograd.grad(ref, [q64, t64], dout.double())
    outs = []
    for _ in range(2):
        qd = qkv.to(DEV).requires_grad_(True)
        td = table.to(DEV).requires_grad_(True)
        y = window_attention_core(qd, td, idx.to(DEV), heads, shift, scale)
        gq, gt = torch.autograd.grad(y, [qd, td], dout.to(DEV))
        outs.append((y.detach(), gq, gt))
    y, gq, gt = outs[0]
    for name, a, r in [("y", y, ref.detach()), ("dqkv", gq, rq), ("dtable", gt, rt)]:
        scale_ = r.abs().max().item()
        err = (a.double().cpu() - r).abs().max().item()
        assert err <= 2e-5 * scale_, (name, err, scale_)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2], outs[1][2]), "bias-table gradient differs between two runs"


def test_foreign_index_layout_is_refused(hip_lib):
    idx = _index().to(DEV)
    idx[0, 1] = 5
    with pytest.raises(NotImplementedError):
        window_attention_core(torch.zeros(1, 7, 7, 12, device=DEV), torch.zeros(169, 2, device=DEV), idx, 2, 0, 1.0)


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_nonfinite_attention_gradient_reaches_the_bias_table_gradient(hip_lib, bad):
    """the bias-table gradient is summed in fixed point (integer LDS adds); a NaN / Inf dS must not come out as a finite
    sum - the fp32 Swin trainers run without a GradScaler, so a non-finite relative_position_bias_table.grad is the only thing
    that flags such a step for this parameter (torch's float sum carries it the same way)."""
    g = torch.Generator().manual_seed(5)
    heads, hd = 3, 32
    qkv = torch.randn(2, 14, 14, 3 * heads * hd, generator=g).to(DEV).requires_grad_(True)
    table = (torch.randn(169, heads, generator=g) * 0.5).to(DEV).requires_grad_(True)
    dout = torch.randn(2, 14, 14, heads * hd, generator=g)
    dout[1, 3, 4, 40] = bad                     # one element of head 1's output gradient in one window
    y = window_attention_core(qkv, table, _index().to(DEV), heads, 3, hd ** -0.5)
    _, gt = torch.autograd.grad(y, [qkv, table], dout.to(DEV))
    assert not torch.isfinite(gt[:, 1]).all()    # head 1's table column carries it
    assert torch.isfinite(gt[:, 0]).all() and torch.isfinite(gt[:, 2]).all()
