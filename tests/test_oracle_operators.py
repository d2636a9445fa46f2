"""Operator-level pinning of the CPU oracles against the reference's own modules (fixtures of tools/make_golden.py): window
attention with / without the cyclic shift (output, dx, every parameter gradient), SURVEY.md 8c KAT-4, a Swin block with the
top / left padding quirk (19 is not a multiple of 7), the SS2D block.  CPU only."""
import os

import numpy as np
import pytest
import torch

from golden_util import det_fill

G = os.path.join(os.path.dirname(__file__), "golden")


def _close(got, ref, rtol, what):
    ref = torch.as_tensor(ref)
    err = (got - ref).abs().max().item()
    assert err <= rtol * ref.abs().max().item() + 1e-7, (what, err, ref.abs().max().item())


@pytest.mark.parametrize("tag", ["s", "n"])
def test_window_attention_equals_the_references(tag):
    from oracle.swt2net import WindowAttention
    z = np.load(os.path.join(G, f"window_attention_{tag}.npz"))
    dim, heads, shift, _ = [int(v) for v in z["cfg"]]
    m = WindowAttention(dim, heads, bool(shift)).eval()
    det_fill(m)
    x = torch.tensor(z["x"]).requires_grad_(True)
    y = m(x)
    _close(y, z["y"], 2e-5, "y")
    params = list(m.named_parameters())
    grads = torch.autograd.grad(y, [x] + [p for _, p in params], torch.tensor(z["dy"]))
    _close(grads[0], z["dx"], 1e-4, "dx")
    for (n, _), g in zip(params, grads[1:]):
        _close(g, z["g_" + n], 1e-4, n)


def test_window_attention_kat4():
    from oracle.swt2net import WindowAttention
    m = WindowAttention(4, 2, True).eval()
    with torch.no_grad():
        for _, p in m.named_parameters():
            p.copy_(torch.linspace(-.5, .5, p.numel()).view_as(p))
        out = m(torch.linspace(-1, 1, 784).view(1, 14, 14, 4))
    _close(out, np.load(os.path.join(G, "window_attention_kat4.npz"))["out"], 2e-5, "kat4")


def test_swin_block_padding_quirk():
    from functools import partial
    from oracle.swt2net import SwinTransformerBlock
    z = np.load(os.path.join(G, "swin_block.npz"))
    blk = SwinTransformerBlock(32, 2, True, 0.0, partial(torch.nn.LayerNorm)).eval()
    det_fill(blk)
    with torch.no_grad():
        _close(blk(torch.tensor(z["x"])), z["y"], 2e-5, "swin block 19x19")


def test_ss2d_block_equals_the_references():
    from oracle.m2net import SS2D
    z = np.load(os.path.join(G, "ss2d.npz"))
    m = SS2D(16).eval()
    det_fill(m)
    names = [n for n, _ in m.named_parameters()]
    assert names == [str(n) for n in z["names"]]
    x = torch.tensor(z["x"]).requires_grad_(True)
    y = m(x)
    _close(y, z["y"], 2e-5, "y")
    grads = torch.autograd.grad(y, [x] + [p for _, p in m.named_parameters()], torch.tensor(z["dy"]))
    for n, g in zip(["dx"] + ["g_" + n for n in names], grads):
        _close(g, z[n], 1e-4, n)
