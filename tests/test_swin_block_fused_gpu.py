"""The fused Swin block (nnuzoo_amd/swin_block.py: five forward / seven backward launches, csrc/dense32.hip prologues and epilogues)
against the module-by-module path of the same SwinTransformerBlock (itself pinned to the reference's block by
tests/golden/swin_block.npz and the whole-net fixtures): same parameters, same input, same DropPath draws - output, dx and every
parameter gradient, inside and outside the grouped weight-gradient pass.  Geometries: the reference's top / left padding quirk
(one axis dividing: a full extra window), no padding, the deep levels whose products go through split-K (8^2 x 768, 16^2 x 384),
a 1 x 1 map (fewer rows per sample than a tile: the per-row DropPath division path)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # B, H, W, C, heads, shift, drop_path
    (2, 9, 11, 32, 2, True, 0.0),
    (2, 14, 21, 64, 4, False, 0.0),      # H divides, W divides: no padding at all
    (2, 14, 16, 64, 2, True, 0.3),       # H divides, W does not: a full extra window of rows
    (3, 16, 16, 96, 3, True, 0.5),
    (2, 8, 8, 768, 24, True, 0.4),       # split-K: fc2 / fc1 dgrad contraction 3072 on 128 tokens
    (2, 16, 16, 384, 12, False, 0.2),
    (4, 1, 1, 128, 4, True, 0.5),
    (2, 32, 32, 128, 8, True, 0.1),
]


def _block(C, heads, shift, dp):
    from nnuzoo_amd.nets.swt2net import SwinTransformerBlock
    torch.manual_seed(7)
    blk = SwinTransformerBlock(C, heads, shift=shift, drop_path=dp).cuda()
    with torch.no_grad():
        for n, p in blk.named_parameters():     # LayerNorm affine and biases away from their 1 / 0 initial values
            if p.dim() == 1:
                p.add_(0.3 * torch.randn_like(p))
        blk.attn.relative_position_bias_table.mul_(25.0)
    return blk


def _run(blk, x, dy, fused, deferred, seed, monkeypatch):
    from nnuzoo_amd.nets import swt2net
    from nnuzoo_amd.token_linear import deferred_wgrads
    monkeypatch.setattr(swt2net, "fused_block_ok", (lambda b, t: True) if fused else (lambda b, t: False))
    for p in blk.parameters():
        p.grad = None
    xx = x.clone().requires_grad_(True)
    torch.manual_seed(seed)
    y = blk(xx)
    if deferred:
        with deferred_wgrads():
            y.backward(dy)
    else:
        y.backward(dy)
    return y.detach(), xx.grad.detach(), {n: p.grad.detach().clone() for n, p in blk.named_parameters()}


@pytest.mark.parametrize("B,H,W,C,heads,shift,dp", CASES)
@pytest.mark.parametrize("deferred", [True, False])
def test_fused_block_equals_module_path(hip_lib, monkeypatch, B, H, W, C, heads, shift, dp, deferred):
    from nnuzoo_amd.swin_block import fused_block_ok
    blk = _block(C, heads, shift, dp).train()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, W, C, generator=g).cuda() * 1.5 + 0.2
    dy = torch.randn(B, H, W, C, generator=g).cuda()
    assert fused_block_ok(blk, x)
    y0, dx0, g0 = _run(blk, x, dy, False, deferred, 11, monkeypatch)
    y1, dx1, g1 = _run(blk, x, dy, True, deferred, 11, monkeypatch)

    def rel(a, b):
        return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)

    assert rel(y1, y0) < 3e-6, rel(y1, y0)
    assert rel(dx1, dx0) < 2e-5, rel(dx1, dx0)
    for n in g0:
        assert rel(g1[n], g0[n]) < 5e-5, (n, rel(g1[n], g0[n]))
    # the fused node is bit-reproducible call to call (fixed split / fold orders everywhere)
    y2, dx2, g2 = _run(blk, x, dy, True, deferred, 11, monkeypatch)
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2) and all(torch.equal(g1[n], g2[n]) for n in g1)


def test_fused_block_eval_mode_and_dropped_samples(hip_lib, monkeypatch):
    """eval: no draws, the DropPath scale is 1; training with a high drop rate: whole samples keep their input unchanged through
    both residuals exactly (x + 0 * branch)"""
    blk = _block(64, 2, True, 0.9).eval()
    x = torch.randn(2, 9, 9, 64, device="cuda")
    y0, _, _ = _run(blk, x, torch.ones_like(x), False, False, 5, monkeypatch)
    y1, _, _ = _run(blk, x, torch.ones_like(x), True, False, 5, monkeypatch)
    assert (y1 - y0).abs().max().item() < 3e-6 * y0.abs().max().item()
    blk.train()
    x = torch.randn(8, 9, 9, 64, device="cuda")
    y1, _, _ = _run(blk, x, torch.ones_like(x), True, False, 5, monkeypatch)
    torch.manual_seed(5)                      # the two draws the block made, attention branch first
    m1 = torch.floor(0.1 + torch.rand((8, 1, 1, 1), device="cuda")).reshape(-1)
    m2 = torch.floor(0.1 + torch.rand((8, 1, 1, 1), device="cuda")).reshape(-1)
    for b in range(8):
        assert y1[b].equal(x[b]) == bool(m1[b] == 0 and m2[b] == 0), (b, m1[b].item(), m2[b].item())


@pytest.mark.parametrize("T,K,N,gelu", [(128, 3072, 768, 0), (392, 768, 3072, 1), (882, 1536, 384, 0), (35378, 32, 96, 0),
                                        (100, 256, 64, 0)])
def test_dense32_fused_entry_points_equal_float64(hip_lib, T, K, N, gelu):
    """nnz_dense32_forward_fused / _dgrad_fused without any fused piece: the plain products, through split-K where the shape asks for
    it, against float64"""
    from nnuzoo_amd._lib import call, ptr, stream_ptr, load
    g = torch.Generator().manual_seed(T + K)
    x = torch.randn(T, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    nws = int(load().nnz_dense32_splitk_workspace_floats(T, K, N))
    ws = torch.empty(max(nws, 1), device="cuda")
    y, ya = torch.empty(T, N, device="cuda"), torch.empty(T, N, device="cuda")
    call("nnz_dense32_forward_fused", ptr(x), ptr(w), ptr(b), ptr(y), ptr(ya) if gelu else None, T, K, N, gelu, None, None, 0.0, None,
         None, None, 0, 0, 0, 0, None, None, 1.0, 1, 1, ptr(ws) if nws else None, stream_ptr())
    ref = x.double() @ w.double().t() + b.double()
    assert (y.double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item() * max(1, K / 512) ** 0.5
    if gelu:
        assert (ya.double() - torch.nn.functional.gelu(ref)).abs().max().item() < 3e-6 * ref.abs().max().item()
    dy = torch.randn(T, N, generator=g).cuda()
    nws = int(load().nnz_dense32_splitk_workspace_floats(T, N, K))
    ws = torch.empty(max(nws, 1), device="cuda")
    dx = torch.empty(T, K, device="cuda")
    call("nnz_dense32_dgrad_fused", ptr(dy), ptr(w), None, ptr(dx), T, K, N, None, 1.0, 1, 1, ptr(ws) if nws else None, stream_ptr())
    dref = dy.double() @ w.double()
    assert (dx.double() - dref).abs().max().item() < 2e-6 * dref.abs().max().item() * max(1, N / 512) ** 0.5
