"""Round 3: deterministic, cancellation-free reductions of the nnU-Net path (csrc/common.hpp FxAcc: fixed-point
cross-workgroup sums, finalised by the launch's last workgroup).

  * InstanceNorm statistics from the conv epilogue / the stats kernel against float64 on the SAME fp16 values at
    |mean| / std = 1 ... 1000 (the fp32 `sumsq / V - mean^2` of rounds 1-2 loses the variance there: VERDICT r2 weak #5);
  * every cross-workgroup sum of the step (statistics, norm backward, stem / head weight gradients, loss sums, gradient
    norm) is order-independent: two runs of the same training steps are BIT-IDENTICAL - outputs, loss, every gradient,
    every parameter after three optimizer steps (rounds 1-2: fp32 atomics made two runs differ by ~5 % in gradient norm
    through LeakyReLU sign flips, tools/probes/run_to_run_first_diff.py);
  * the scratch accumulators are left zero by every launch.
Reference semantics: nn.InstanceNorm3d(eps 1e-5, affine) biased variance (default_experiment_planner.py:285-305)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from nnuzoo_amd import conv_plan as cp
from nnuzoo_amd import hip_ops as ops
from nnuzoo_amd.hip_ops import PreparedTable

DEV = "cuda"


def _ref_table(raw16: torch.Tensor, gamma, beta, eps):
    x = raw16.double().cpu()                       # [N, V, C]
    mean = x.mean(1)
    var = x.var(1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + eps)
    scale = rstd * gamma.double().cpu()
    return torch.stack([mean, rstd, scale, beta.double().cpu() - mean * scale], dim=-1)


def _check_table(nstat, ref, what, rstd_tol=3e-6):
    got = nstat.double().cpu()
    for k, name in enumerate(["mean", "rstd", "scale", "shift"]):
        r = ref[..., k]
        # shift = beta - mean * scale inherits rstd's relative error times |mean * scale|
        scale = (ref[..., 0] * ref[..., 2]).abs().max().item() + 1.0 if name == "shift" else r.abs().max().item()
        err = (got[..., k] - r).abs().max().item()
        assert err <= (3e-6 if name == "mean" else rstd_tol) * scale + 1e-7, (what, name, err, scale)


@pytest.mark.parametrize("ratio", [1.0, 100.0, 1000.0])
@pytest.mark.parametrize("N,V,C", [(2, 40000, 32), (1, 777, 64), (3, 5000, 320)])
def test_stats_kernel_large_mean(hip_lib, ratio, N, V, C):
    g = torch.Generator().manual_seed(int(ratio) + C)
    std = 0.05
    x = (ratio * std * (1 + 0.1 * torch.randn(1, 1, C, generator=g)) + std * torch.randn(N, V, C, generator=g))
    x16 = x.to(torch.float16).to(DEV)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    sc = ops.NormScratch(torch.device(DEV), N * C)
    nstat = torch.full((N, C, 4), float("nan"), device=DEV)
    sums = torch.full((N, C, 2), float("nan"), device=DEV)
    ops.instnorm_stats_det(x16, N, V, C, C, sc, gamma, beta, 1e-5, nstat=nstat, sums=sums)
    torch.cuda.synchronize()
    _check_table(nstat, _ref_table(x16, gamma, beta, 1e-5), f"stats kernel ratio {ratio}")
    x64 = x16.double().cpu()
    assert torch.allclose(sums[..., 0].double().cpu(), x64.sum(1), rtol=1e-6)
    assert torch.allclose(sums[..., 1].double().cpu(), (x64 * x64).sum(1), rtol=1e-6)
    assert int(sc.acc.abs().max()) == 0 and int(sc.counter.abs().max()) == 0     # left ready for the next launch
    # and the apply / backward kernels on that table against float64 autograd on the same fp16 values
    y = torch.empty_like(x16)
    ops.instnorm_lrelu_apply_tab(x16, nstat, y, N, V, C, C, C, 0.01)
    xr = x16.double().cpu().requires_grad_(True)
    mu = xr.mean(1, keepdim=True)
    xn = (xr - mu) / torch.sqrt(xr.var(1, unbiased=False, keepdim=True) + 1e-5)
    pre = xn * gamma.double().cpu() + beta.double().cpu()
    yr = torch.nn.functional.leaky_relu(pre, 0.01)
    assert (y.double().cpu() - yr.detach()).abs().max().item() <= 2e-3 * yr.abs().max().item() + 2e-3
    gy = torch.randn(N, V, C, generator=g).to(torch.float16)
    yr.backward(gy.double())
    nred = torch.empty((N, C, 2), device=DEV)
    dx = torch.empty_like(x16)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.instnorm_lrelu_bwd_tab(x16, gy.to(DEV), nstat, sc, nred, dx, N, V, C, C, C, C, 0.01, dgamma=dg, dbeta=db)
    torch.cuda.synchronize()
    if ratio <= 100:     # beyond that the fp16 grid of x itself is as coarse as std: sign(pre) of single voxels differs
        ref_dx = xr.grad
        assert (dx.double().cpu() - ref_dx).abs().max().item() <= 5e-3 * ref_dx.abs().max().item()
    assert int(sc.acc.abs().max()) == 0 and int(sc.counter.abs().max()) == 0


@pytest.mark.parametrize("dims,cin,cout,stride", [((16, 16, 16), 32, 32, 1), ((64, 64, 64), 32, 32, 1),
                                                   ((16, 24, 8), 64, 64, 1), ((16, 16, 16), 32, 64, 2),
                                                   ((1, 64, 48), 32, 32, 1), ((12, 20, 28), 32, 64, 1)])
@pytest.mark.parametrize("mfma_moments", [1, 0])
def test_conv_epilogue_table_large_mean(hip_lib, dims, cin, cout, stride, mfma_moments):
    """the forward convolution's own statistics (all tile shapes incl. ragged edges and the depth-reuse loop) with a bias
    that puts |mean| at 20 ... 200 std: table vs float64 on the stored fp16 outputs; outputs identical to the plain launch.
    Both forms of the tile moments: on the matrix cores for tiles inside the volume (nnz_conv_tuning knob 11, the default) and the
    VALU sums everywhere."""
    from nnuzoo_amd import _lib
    _lib.call("nnz_conv_tuning", 11, mfma_moments)
    try:
        _conv_epilogue_table_large_mean(dims, cin, cout, stride)
    finally:
        _lib.call("nnz_conv_tuning", 11, 1)


@pytest.mark.parametrize("dims,cin,cout", [((32, 32, 32), 32, 32), ((16, 16, 16), 256, 256), ((8, 8, 8), 256, 256),
                                           ((12, 20, 28), 32, 64)])
@pytest.mark.parametrize("mfma_moments", [1, 0])
def test_conv_epilogue_table_small_mean(hip_lib, dims, cin, cout, mfma_moments):
    """the opposite regime: channel means within a standard deviation of zero, where the pilot subtraction of the matrix-core
    moments (knob 11) ROUNDS in fp16.  The table's mean must stay exact in both forms (it comes from the exact sum of x); the
    variance of the matrix-core form is the variance of the fp16-rounded deviations: rstd within 1e-4 relative (the VALU form:
    3e-6)."""
    from nnuzoo_amd import _lib
    _lib.call("nnz_conv_tuning", 11, mfma_moments)
    try:
        _conv_epilogue_table_large_mean(dims, cin, cout, 1, bias_mean=0.0, w_scale=0.05, rstd_tol=1e-4 if mfma_moments else 3e-6)
    finally:
        _lib.call("nnz_conv_tuning", 11, 1)


def _conv_epilogue_table_large_mean(dims, cin, cout, stride, bias_mean=6.0, w_scale=0.002, rstd_tol=3e-6):
    N = 2
    g = torch.Generator().manual_seed(sum(dims) + cout)
    x = torch.randn(N, int(np.prod(dims)), cin, generator=g).to(torch.float16).to(DEV)
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) * w_scale).to(DEV)
    ks = (1, 3, 3) if dims[0] == 1 else (3, 3, 3)
    if dims[0] == 1:
        w = w[:, :, 1:2].contiguous()
    b = (bias_mean + 0.5 * torch.randn(cout, generator=g)).to(DEV)
    st = (1, stride, stride) if dims[0] == 1 else stride
    pt = PreparedTable(cp.conv_forward(N, dims, cin, cout, ks=ks, stride=st))
    nk = int(np.prod(ks))
    wp = ops.pack_weight(w, pt, cin, cout, nk, cin * nk, 1)
    odims = cp.conv_out_dims(dims, ks, (st,) * 3 if isinstance(st, int) else st)
    V = int(np.prod(odims))
    out0 = torch.empty((N, V, cout), dtype=torch.float16, device=DEV)
    ops.conv_tap_forward(pt, x, wp, b, out0)
    gamma = (1 + 0.1 * torch.randn(cout, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(cout, generator=g)).to(DEV)
    sc = ops.NormScratch(torch.device(DEV), N * cout)
    for rep in range(2):                          # the second launch finds the accumulators the first one left
        out = torch.empty_like(out0)
        nstat = torch.full((N, cout, 4), float("nan"), device=DEV)
        ops.conv_tap_forward_norm(pt, x, wp, b, out, sc, gamma, beta, 1e-5, nstat)
        torch.cuda.synchronize()
        assert torch.equal(out, out0)
        ratio = (out.double().mean(1).abs() / out.double().std(1)).min().item()
        assert ratio > 20 or bias_mean == 0.0, ratio
        _check_table(nstat, _ref_table(out, gamma, beta, 1e-5), f"conv epilogue rep {rep}", rstd_tol)
        assert int(sc.acc.abs().max()) == 0 and int(sc.counter.abs().max()) == 0


def _three_steps(seed: int):
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer
    plans, cfg, dj = nnunet_plans(3, (64, 64, 64), batch_size=2)
    torch.manual_seed(seed)
    tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device(DEV))
    tr.initialize()
    b = synthetic_batch(2, (64, 64, 64), tr._get_deep_supervision_scales(), seed=3)
    b = {"data": b["data"].to(DEV), "target": [t.to(DEV) for t in b["target"]]}
    losses, grads = [], None
    for i in range(3):
        losses.append(float(tr.train_step(b)["loss"]))
        if i == 0:
            grads = tr.network.grad_arena().clone()
    with torch.no_grad():
        tr.network.eval()
        out = [o.clone() for o in tr.network(b["data"])]
    return losses, grads, [p.detach().clone() for p in tr.network.parameters()], out


def test_training_steps_are_bit_identical_run_to_run(hip_lib):
    a = _three_steps(0)
    junk = [torch.randn(1 + 977 * i, device=DEV) for i in range(64)]      # different allocator state / timing
    b = _three_steps(0)
    del junk
    assert a[0] == b[0], (a[0], b[0])                                      # losses of the three steps
    assert torch.equal(a[1], b[1])                                          # the whole gradient arena of step 0
    assert all(torch.equal(p, q) for p, q in zip(a[2], b[2]))               # parameters after three SGD steps
    assert all(torch.equal(p, q) for p, q in zip(a[3], b[3]))               # logits of the trained network
    assert float(a[1].abs().max()) > 0 and all(np.isfinite(a[0]))


@pytest.mark.parametrize("dims,cin,cout,stride,acc", [((16, 16, 16), 32, 32, 1, False), ((32, 32, 32), 32, 64, 1, False),
                                                       ((64, 64, 64), 32, 32, 1, False), ((16, 16, 16), 32, 64, 2, True),
                                                       ((24, 16, 8), 64, 128, 2, True), ((12, 20, 28), 64, 64, 1, False),
                                                       ((8, 8, 8), 256, 320, 2, True), ((8, 8, 8), 320, 320, 1, False),
                                                       ((1, 32, 48), 32, 32, 1, False), ((1, 32, 48), 32, 64, 2, True)])
def test_dgrad_epilogue_closes_norm_backward_reductions(hip_lib, dims, cin, cout, stride, acc):
    """nnz_conv_tap_dgrad_normred: the data-gradient launch of conv(cin -> cout) also forms {mean g', mean g' xhat}, dgamma,
    dbeta of the InstanceNorm + LeakyReLU below it (cin channels) - against the separate reducing launch on the same g and
    against float64; g itself is bit-identical to the plain launch (stride-2 phase groups, accumulate into a 2C-strided
    skip slice, ragged tiles, the 2-D tables, the split-K-sized 8^3 levels)."""
    N = 2
    g = torch.Generator().manual_seed(sum(dims) + cout + stride)
    flat = dims[0] == 1
    ks = (1, 3, 3) if flat else (3, 3, 3)
    st = ((1, stride, stride) if flat else (stride,) * 3)
    ldo = 2 * cin if acc else cin
    pt = PreparedTable(cp.conv_dgrad(N, dims, cin, cout, ks=ks, stride=st, ldi=cout, ldo=ldo, accumulate=acc))
    nk = int(np.prod(ks))
    w = (torch.randn(cout, cin, *ks, generator=g) * 0.05).to(DEV)
    wp = ops.pack_weight(w, pt, cout, cin, 1, cin * nk, nk)         # any packing serves both launches alike
    ydims = cp.conv_out_dims(dims, ks, st)
    dy = torch.randn(N, int(np.prod(ydims)), cout, generator=g).to(torch.float16).to(DEV)
    V = int(np.prod(dims))
    base = (torch.randn(N, V, ldo, generator=g) * 0.5).to(torch.float16).to(DEV)
    x_raw = (0.3 + torch.randn(N, V, cin, generator=g)).to(torch.float16).to(DEV)
    gamma = (1 + 0.1 * torch.randn(cin, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(cin, generator=g)).to(DEV)
    sc = ops.NormScratch(torch.device(DEV), N * max(cin, cout))
    nstat = torch.empty((N, cin, 4), device=DEV)
    ops.instnorm_stats_det(x_raw, N, V, cin, cin, sc, gamma, beta, 1e-5, nstat=nstat)
    # separate launches
    out0 = base.clone()
    view0 = out0[:, :, ldo - cin:]
    ops.conv_tap_forward(pt, dy, wp, None, view0)
    nred0 = torch.empty((N, cin, 2), device=DEV)
    dg0, db0 = torch.empty(cin, device=DEV), torch.empty(cin, device=DEV)
    dx0 = torch.empty((N, V, cin), dtype=torch.float16, device=DEV)
    ops.instnorm_lrelu_bwd_tab(x_raw, view0, nstat, sc, nred0, dx0, N, V, cin, cin, ldo, cin, 0.01, dgamma=dg0, dbeta=db0)
    res = []
    for _ in range(2):
        out1 = base.clone()
        view1 = out1[:, :, ldo - cin:]
        nred1 = torch.full((N, cin, 2), float("nan"), device=DEV)
        dg1, db1 = torch.full((cin,), float("nan"), device=DEV), torch.full((cin,), float("nan"), device=DEV)
        ops.conv_tap_dgrad_normred(pt, dy, wp, view1, x_raw, cin, nstat, 0.01, sc, nred1, dg1, db1)
        dx1 = torch.empty((N, V, cin), dtype=torch.float16, device=DEV)
        ops.instnorm_lrelu_bwd_apply_tab(x_raw, view1, nstat, nred1, dx1, N, V, cin, cin, ldo, cin, 0.01)
        torch.cuda.synchronize()
        assert int(sc.acc.abs().max()) == 0 and int(sc.counter.abs().max()) == 0
        assert torch.equal(out1, out0)                 # g (and the untouched half of a skip buffer) bit for bit
        res.append((nred1, dg1, db1, dx1))
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    # float64 on the stored fp16 g
    gg = view0.double().cpu()
    xx = x_raw.double().cpu()
    t = nstat.double().cpu()
    pre = xx * t[:, None, :, 2] + t[:, None, :, 3]
    gp = torch.where(pre > 0, gg, gg * 0.01)
    xn = (xx - t[:, None, :, 0]) * t[:, None, :, 1]
    ref = torch.stack([gp.mean(1), (gp * xn).mean(1)], -1)
    for name, got, sep, r in [("nred", res[0][0], nred0, ref), ("dgamma", res[0][1], dg0, (gp * xn).sum((0, 1))),
                              ("dbeta", res[0][2], db0, gp.sum((0, 1)))]:
        scale = r.abs().max().item()
        noise = (gp.abs().mean().item() * (V ** -0.5 if name == "nred" else (N * V) ** 0.5)) * 2e-6 + 1e-6 * scale
        assert (got.double().cpu() - r).abs().max().item() <= noise + 2e-6 * scale, name
        assert (sep.double().cpu() - r).abs().max().item() <= noise + 2e-6 * scale, name + " (separate launch)"
    assert (res[0][3].float() - dx0.float()).abs().max().item() <= 2e-3 * dx0.float().abs().max().item()
