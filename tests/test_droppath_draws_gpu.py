"""nnuzoo_amd/droppath_draws.py: every stochastic-depth draw of a forward pass from one `torch.rand` launch (~240
`torch.rand` launches per SwT2Net step).  Checked: the first training pass counts the requests, later passes serve them from the
table; equal seeds give equal passes, different seeds different ones; the masks are Bernoulli(keep) per sample; eval passes draw
nothing."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(name):
    from nnuzoo_amd.nets import m2net, swt2net
    torch.manual_seed(0)
    return {"M2NetP": m2net.M2NetP, "SwT2Net": swt2net.SwT2Net}[name](1, 2, True).cuda().train()


@pytest.mark.parametrize("name", ["M2NetP", "SwT2Net"])
def test_draws_come_from_one_table_per_pass(hip_lib, name):
    from nnuzoo_amd import droppath_draws as dd
    net = _net(name)
    x = torch.randn(2, 1, 64, 64, generator=torch.Generator().manual_seed(1)).cuda()
    calls = []
    real = torch.rand

    def counting(*a, **k):
        calls.append(a[0] if a else None)
        return real(*a, **k)

    torch.rand = counting
    try:
        with torch.no_grad():
            net(x)                                       # learning pass: per-request calls
            n_requests = net._droppath_requests
            first = len(calls)
            calls.clear()
            torch.manual_seed(7)
            a = [o.clone() for o in net(x)]
            table_calls = list(calls)
            torch.manual_seed(7)
            b = [o.clone() for o in net(x)]
            torch.manual_seed(8)
            c = [o.clone() for o in net(x)]
    finally:
        torch.rand = real
    assert n_requests > 10 and first >= n_requests       # every stochastic block asked once per branch
    # ... and the second pass made ONE call for all of them (plus whatever calls did not go through the table in the first pass)
    assert table_calls.count((n_requests, 2)) == 1 and len(table_calls) == 1 + (first - n_requests), (table_calls[:4], first)
    # equal seeds -> equal masks -> equal outputs up to the library convolutions' run-to-run noise; another seed drops other blocks
    def near(p, q):
        return (p - q).abs().max().item() <= 1e-3 * p.abs().max().item()
    assert all(near(p, q) for p, q in zip(a, b)), [(p - q).abs().max().item() / p.abs().max().item() for p, q in zip(a, b)]
    assert any(not near(p, q) for p, q in zip(a, c))
    net.eval()
    calls.clear()
    torch.rand = counting
    try:
        with torch.no_grad():
            net(x)
    finally:
        torch.rand = real
    assert calls == [] and net._droppath_requests == n_requests
    assert not dd._ACTIVE


def test_mask_is_bernoulli_keep(hip_lib):
    """floor(keep + u) of the table's uniform rows inside the residual kernel: the kept fraction over many samples is `keep`, kept
    samples are scaled by 1 / keep"""
    from nnuzoo_amd.droppath_draws import DrawTable
    from nnuzoo_amd.nets.common2d import DropPath, residual_drop_path

    class Owner(torch.nn.Module):
        pass

    torch.manual_seed(0)
    own = Owner().train()
    dp = DropPath(0.3).train()
    B = 4096
    inp = torch.zeros(B, 8, device="cuda")
    x = torch.ones(B, 8, device="cuda")
    for _ in range(2):                                   # the second pass is served by the table
        with DrawTable(own, B, inp.device):
            out = residual_drop_path(inp, x, dp)
    assert own._droppath_requests == 1
    kept = (out[:, 0] != 0).float().mean().item()
    assert abs(kept - 0.7) < 0.03, kept
    vals = out[out[:, 0] != 0]
    assert torch.allclose(vals, torch.full_like(vals, 1 / 0.7), rtol=1e-6)
