"""GPU parity of the HIP selective scan (fwd + 7 gradients, through the C-ABI) against
  (1) the golden vectors produced by the reference's own selective_scan_ref (tests/golden/selective_scan_*.npz),
  (2) the CPU oracle on fresh random inputs, including multi-chunk L and L not divisible by 4,
  (3) at the benchmark's full sequence length (L = 262144) a size-independent property: the scan of a
      concatenation equals scan(first half) continued by scan(second half) -> checked through chunk-boundary
      consistency against the oracle on a strided subsample of rows, plus linearity in u.
Tolerance: fp32, chunk-parallel vs sequential summation order -> rtol 1e-4 / atol 1e-4 * max|ref|."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.selective_scan import selective_scan_torch
from nnuzoo_amd.selective_scan import selective_scan_fn

G = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["u", "delta", "A", "B", "C", "D", "delta_bias"]


def close(got, ref, name, rtol=1e-4):
    ref = torch.as_tensor(ref)
    atol = rtol * max(ref.abs().max().item(), 1e-6)
    assert torch.allclose(got.cpu(), ref, rtol=rtol, atol=atol), \
        (name, (got.cpu() - ref).abs().max().item(), ref.abs().max().item())


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "selective_scan_[a-d].npz"))))
def test_against_reference_golden(hip_lib, path):
    z = np.load(path)
    t = {k: torch.tensor(z[k]).cuda().requires_grad_(True) for k in NAMES}
    y = selective_scan_fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], None, t["delta_bias"], True)
    close(y.detach(), z["y"], "y")
    grads = torch.autograd.grad(y, [t[k] for k in NAMES], torch.tensor(z["dy"]).cuda())
    for n, g in zip(["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"], grads):
        close(g, z[n], n, rtol=2e-4)


def test_kat1_shape_rejected_loudly(hip_lib):
    # KAT-1 uses d_state = 2; the kernels are specialised to the zoo's d_state = 16 and must say so
    with pytest.raises(NotImplementedError):
        selective_scan_fn(torch.zeros(1, 2, 4).cuda(), torch.zeros(1, 2, 4).cuda(), torch.zeros(2, 2).cuda(),
                          torch.zeros(1, 1, 2, 4).cuda(), torch.zeros(1, 1, 2, 4).cuda())


@pytest.mark.parametrize("b,K,Dg,L", [(2, 4, 32, 1024), (1, 4, 8, 777), (2, 6, 4, 259), (1, 1, 64, 2048)])
def test_against_oracle(hip_lib, b, K, Dg, L):
    g = torch.Generator().manual_seed(L)
    KD, N = K * Dg, 16
    inp = dict(u=torch.randn(b, KD, L, generator=g), delta=torch.randn(b, KD, L, generator=g) * 0.5,
               A=-torch.exp(torch.randn(KD, N, generator=g) * 0.5), B=torch.randn(b, K, N, L, generator=g),
               C=torch.randn(b, K, N, L, generator=g), D=torch.randn(KD, generator=g),
               delta_bias=torch.randn(KD, generator=g) * 0.5 - 1)
    ref_in = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
    yr = selective_scan_torch(*[ref_in[k] for k in ["u", "delta", "A", "B", "C", "D", "delta_bias"]], True)
    dy = torch.randn(yr.shape, generator=g)
    gr = torch.autograd.grad(yr, [ref_in[k] for k in NAMES], dy)
    t = {k: v.cuda().requires_grad_(True) for k, v in inp.items()}
    y = selective_scan_fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], None, t["delta_bias"], True)
    close(y.detach(), yr.detach(), "y")
    gg = torch.autograd.grad(y, [t[k] for k in NAMES], dy.cuda())
    for n, a, r in zip(NAMES, gg, gr):
        close(a, r, "d" + n, rtol=3e-4)


@pytest.mark.parametrize("b,K,Dg,L,clb,opt", [(2, 4, 32, 1024, 0, True), (1, 4, 32, 192, 4, True), (1, 1, 64, 2048, 16, True),
                                              (2, 6, 64, 256, 4, False), (1, 2, 128, 512, 8, True),
                                              (1, 1, 192, 1024, 64, False)])
def test_against_oracle_channels_on_lanes(hip_lib, b, K, Dg, L, clb, opt):
    """the second-generation kernels (csrc/ss2d_scan_rl.hpp, plain-API instantiation; by default only for >= 2 M row-steps)
    forced on for small shapes: two chunk slots per wave (Dg = 32, incl. an odd chunk count = idle slot), one slot
    (Dg = 64), several channel groups adding into one dB / dC tile (Dg = 128, 192), chunk lengths 64..1024 steps, with
    and without D / delta_bias / softplus - against the fp32 CPU oracle (oracle/selective_scan.py: the reference's sequential recurrence)"""
    from nnuzoo_amd._lib import call, load
    lib = load()
    g = torch.Generator().manual_seed(L + Dg)
    KD, N = K * Dg, 16
    inp = dict(u=torch.randn(b, KD, L, generator=g), delta=torch.randn(b, KD, L, generator=g) * 0.5,
               A=-torch.exp(torch.randn(KD, N, generator=g) * 0.5), B=torch.randn(b, K, N, L, generator=g),
               C=torch.randn(b, K, N, L, generator=g), D=torch.randn(KD, generator=g),
               delta_bias=torch.randn(KD, generator=g) * 0.5 - 1)
    if not opt:
        inp["delta"] = inp["delta"].abs() * 0.2 + 0.01
    names = NAMES if opt else NAMES[:5]
    ref_in = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
    args = lambda d: [d[k] for k in NAMES[:5]] + ([d["D"], None, d["delta_bias"], True] if opt else [None, None, None, False])
    yr = selective_scan_torch(*[ref_in[k] for k in NAMES[:5]],
                              *([ref_in["D"], ref_in["delta_bias"], True] if opt else [None, None, False]))
    dy = torch.randn(yr.shape, generator=g)
    gr = torch.autograd.grad(yr, [ref_in[k] for k in names], dy)
    t = {k: v.cuda().requires_grad_(True) for k, v in inp.items()}
    saved = [lib.nnz_scan_tuning_get(k) for k in range(3)]
    try:
        call("nnz_scan_tuning", 0, 1)
        call("nnz_scan_tuning", 1, clb)
        call("nnz_scan_tuning", 2, 0)
        before = lib.nnz_scan_tuning_get(3)
        y = selective_scan_fn(*args(t))
        gg = torch.autograd.grad(y, [t[k] for k in names], dy.cuda())
        assert lib.nnz_scan_tuning_get(3) == before + 2          # forward and backward both took the new kernels
    finally:
        for k, v in enumerate(saved):
            call("nnz_scan_tuning", k, v)
    close(y.detach(), yr.detach(), "y")
    for n, a, r in zip(names, gg, gr):
        close(a, r, "d" + n, rtol=3e-4)


def test_full_length_properties(hip_lib):
    """L = 512*512 as in the first SS2D block of M2Net at 512^2 (u: (1, 128, 262144))."""
    g = torch.Generator().manual_seed(0)
    b, K, Dg, N, L = 1, 4, 32, 16, 512 * 512
    KD = K * Dg
    u = torch.randn(b, KD, L, generator=g).cuda()
    delta = (torch.randn(b, KD, L, generator=g) * 0.5).cuda()
    A = -torch.exp(torch.randn(KD, N, generator=g) * 0.3).cuda()
    B = torch.randn(b, K, N, L, generator=g).cuda()
    C = torch.randn(b, K, N, L, generator=g).cuda()
    bias = (torch.randn(KD, generator=g) * 0.5 - 1).cuda()
    y = selective_scan_fn(u, delta, A, B, C, None, None, bias, True)
    # (i) linearity in u (the recurrence is linear in u for fixed delta, B, C)
    y2 = selective_scan_fn(2.5 * u, delta, A, B, C, None, None, bias, True)
    assert torch.allclose(y2, 2.5 * y, rtol=1e-4, atol=1e-4 * y.abs().max().item())
    # (ii) prefix consistency: the first 3000 steps do not depend on the rest (causality + chunk carries)
    Lp = 3000
    yp = selective_scan_fn(u[..., :Lp].contiguous(), delta[..., :Lp].contiguous(), A, B[..., :Lp].contiguous(),
                           C[..., :Lp].contiguous(), None, None, bias, True)
    assert torch.allclose(yp, y[..., :Lp], rtol=1e-4, atol=1e-4 * y.abs().max().item())
    # (iii) exact oracle on the LAST 2000 steps of one row, restarted from a zero state 6000 steps earlier:
    #       exp(delta*A) <= exp(-softplus(.)*|A|) decays any earlier state far below fp32 resolution
    r, t0, t1 = 5, L - 8000, L - 2000
    yo = selective_scan_torch(u[:, r:r + 1, t0:].cpu(), delta[:, r:r + 1, t0:].cpu(), A[r:r + 1].cpu(),
                              B[:, 0:1, :, t0:].cpu(), C[:, 0:1, :, t0:].cpu(), None, bias[r:r + 1].cpu(), True)
    got = y[:, r:r + 1, t1:].cpu()
    assert torch.allclose(got, yo[..., t1 - t0:], rtol=2e-4, atol=2e-4 * yo.abs().max().item())


def _full_inputs(seed=0, b=1, K=4, Dg=32, L=512 * 512):
    g = torch.Generator().manual_seed(seed)
    KD, N = K * Dg, 16
    t = dict(u=torch.randn(b, KD, L, generator=g), delta=torch.randn(b, KD, L, generator=g) * 0.5,
             A=-torch.exp(torch.randn(KD, N, generator=g) * 0.3), B=torch.randn(b, K, N, L, generator=g),
             C=torch.randn(b, K, N, L, generator=g), D=torch.randn(KD, generator=g),
             delta_bias=torch.randn(KD, generator=g) * 0.5 - 1)
    return {k: v.cuda() for k, v in t.items()}


def _scan_grads(t, dy, gen, force_scan_gen2):
    """y and the 7 gradients on generation `gen` (1: time on lanes, 2: channels on lanes) at any size"""
    from nnuzoo_amd._lib import call, load
    lib = load()
    leaf = {k: v.clone().requires_grad_(True) for k, v in t.items()}
    run = lambda: selective_scan_fn(leaf["u"], leaf["delta"], leaf["A"], leaf["B"], leaf["C"], leaf["D"], None,
                                    leaf["delta_bias"], True)
    if gen == 2:
        with force_scan_gen2(0) as taken:
            y = run()
            grads = torch.autograd.grad(y, [leaf[k] for k in NAMES], dy)
            assert taken() == 2
    else:
        saved = lib.nnz_scan_tuning_get(0)
        call("nnz_scan_tuning", 0, 0)
        try:
            before = lib.nnz_scan_tuning_get(3)
            y = run()
            grads = torch.autograd.grad(y, [leaf[k] for k in NAMES], dy)
            assert lib.nnz_scan_tuning_get(3) == before
        finally:
            call("nnz_scan_tuning", 0, saved)
    return y.detach(), dict(zip(NAMES, grads))


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_full_length_backward_properties(hip_lib, force_scan_gen2):
    """The BACKWARD at the bench's sequence length, (1, 128, 262144) - checkpoint replay across hundreds of chunks,
    column-sum reductions of dB / dC, the reverse carry: size-independent properties, since the sequential oracle cannot
    run 262 144 steps in test time.
      (i)   generation 2 (default at this size) and generation 1 - two unrelated kernel designs - agree on y and on all
            7 gradients;
      (ii)  every gradient is linear in dy (fixed forward operands);
      (iii) SUFFIX causality: gradients at steps >= t0 depend only on dy at steps >= t0 - with dy zeroed in front of
            t0, du / ddelta / dB / dC on the tail equal those of the full dy exactly-to-rounding, and are zero in front of
            the window the decay allows;
      (iv)  the oracle's own autograd on the LAST 6000 steps of one channel group restarted from a zero state (the
            state entering the window has decayed below fp32 resolution by then for the forward; the backward's
            reverse-time adjoint starts at the sequence end, so the tail is its exact initial segment): du, ddelta, dB,
            dC on the final 2000 steps."""
    t = _full_inputs()
    L = t["u"].shape[-1]
    g = torch.Generator().manual_seed(1)
    dy = torch.randn(t["u"].shape, generator=g).cuda()
    y2, g2 = _scan_grads(t, dy, 2, force_scan_gen2)
    y1, g1 = _scan_grads(t, dy, 1, force_scan_gen2)
    assert _rel(y2, y1) < 1e-4
    for n in NAMES:
        assert _rel(g2[n], g1[n]) < (2e-3 if n in ("A", "D", "delta_bias") else 3e-4), (n, _rel(g2[n], g1[n]))
    # (ii) linearity in dy, on the default (generation 2) path
    _, g2s = _scan_grads(t, -1.75 * dy, 2, force_scan_gen2)
    for n in NAMES:
        assert _rel(g2s[n], -1.75 * g2[n]) < 2e-4, n
    # (iii) suffix causality
    t0 = L - 70000
    dyz = dy.clone()
    dyz[..., :t0] = 0
    _, gz = _scan_grads(t, dyz, 2, force_scan_gen2)
    for n in ("u", "delta", "B", "C"):
        a, b = gz[n][..., t0:], g2[n][..., t0:]
        assert _rel(a, b) < 2e-4, (n, _rel(a, b))
    assert float(gz["C"][..., :t0].abs().max()) == 0.0          # dC_t = dy_t * h_t: no dy, no gradient
    assert float(gz["u"][..., :t0 - 20000].abs().max()) < 1e-6 * float(g2["u"].abs().max())
    # (iv) oracle autograd on the tail of one (batch, direction) group: rows of group k = 1, the first 4 channels
    k, rows, w0, w1 = 1, slice(32, 36), L - 6000, L - 2000
    leaf = dict(u=t["u"][:, rows, w0:].cpu(), delta=t["delta"][:, rows, w0:].cpu(), A=t["A"][rows].cpu(),
                B=t["B"][:, k:k + 1, :, w0:].cpu(), C=t["C"][:, k:k + 1, :, w0:].cpu(), D=t["D"][rows].cpu(),
                delta_bias=t["delta_bias"][rows].cpu())
    leaf = {n: v.clone().requires_grad_(True) for n, v in leaf.items()}
    yo = selective_scan_torch(*[leaf[n] for n in NAMES], True)
    assert _rel(y2[:, rows, w1:].cpu(), yo[..., w1 - w0:].detach()) < 2e-4
    go = torch.autograd.grad(yo, [leaf["u"], leaf["delta"]], dy[:, rows, w0:].cpu())
    assert _rel(g2["u"][:, rows, w1:].cpu(), go[0][..., w1 - w0:]) < 3e-4
    assert _rel(g2["delta"][:, rows, w1:].cpu(), go[1][..., w1 - w0:]) < 3e-4
    # dB / dC sum over the 32 channels of the group: the oracle runs the whole group on the final window only
    rows = slice(32, 64)
    leaf = dict(u=t["u"][:, rows, w0:].cpu(), delta=t["delta"][:, rows, w0:].cpu(), A=t["A"][rows].cpu(),
                B=t["B"][:, k:k + 1, :, w0:].cpu(), C=t["C"][:, k:k + 1, :, w0:].cpu(), D=t["D"][rows].cpu(),
                delta_bias=t["delta_bias"][rows].cpu())
    leaf = {n: v.clone().requires_grad_(True) for n, v in leaf.items()}
    yo = selective_scan_torch(*[leaf[n] for n in NAMES], True)
    go = torch.autograd.grad(yo, [leaf["B"], leaf["C"]], dy[:, rows, w0:].cpu())
    assert _rel(g2["B"][:, k:k + 1, :, w1:].cpu(), go[0][..., w1 - w0:]) < 3e-4
    assert _rel(g2["C"][:, k:k + 1, :, w1:].cpu(), go[1][..., w1 - w0:]) < 3e-4
