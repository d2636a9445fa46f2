"""The fused SS2D core (nnuzoo_amd/ss2d_scan.py: cross-scan mode of the chunk-scan kernels + layout kernels).

Two kinds of checks live here:
  * PARITY (round 6): the fused HIP block against the CPU ORACLE's SS2D (oracle/m2net.py, the plain-torch restatement of
    /root/reference/nnunetv2/nets/m2net.py:39-206 that tests/test_oracle_operators.py pins to the reference's own module output on
    tests/golden/ss2d.npz) on fresh seeded inputs - output, dx and every parameter gradient (test_fused_block_matches_the_cpu_oracle);
  * REGRESSION: the fused formulation against the op-by-op formulation of the reference's SS2D.forward_core around selective_scan_fn
    (SS2D.fused_cross_scan = False; both are HIP paths - a self-consistency check, not parity; the op-by-op path itself is pinned to
    the reference through test_zoo_gpu.py / test_selective_scan_gpu.py).  fp32: outputs rtol 1e-4, gradients rtol 2e-3 of the
    largest entry.  Since round 5 neither formulation uses float atomics: the backward reduces through per-workgroup slabs and a
    fixed-order fold (bit-reproducible, test_scan_backward_is_bit_reproducible...)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(block, x, dy, fused, autocast=False, dwconv=True):
    from nnuzoo_amd.nets.m2net import SS2D
    block.zero_grad(set_to_none=True)
    old, old_dw = SS2D.fused_cross_scan, SS2D.fused_dwconv
    SS2D.fused_cross_scan, SS2D.fused_dwconv = fused, dwconv
    try:
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
            y = block(xi)
        y.float().backward(dy)
    finally:
        SS2D.fused_cross_scan, SS2D.fused_dwconv = old, old_dw
    grads = {n: p.grad.clone() for n, p in block.named_parameters() if p.grad is not None}
    return y.detach().float(), xi.grad.clone(), grads


def _close(a, b, tol, what):
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    assert err <= tol * scale, (what, err, scale)


@pytest.mark.parametrize("d_model,B,H,W", [(16, 2, 24, 40), (16, 1, 13, 9), (32, 2, 16, 16), (64, 1, 20, 12),
                                           (128, 2, 8, 8), (16, 1, 64, 48), (16, 2, 128, 64)])
def test_fused_core_matches_composed_fp32(hip_lib, d_model, B, H, W):
    from nnuzoo_amd.nets.m2net import SS2D
    torch.manual_seed(d_model + H)
    blk = SS2D(d_model=d_model).cuda()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, d_model, generator=g).cuda()
    dy = torch.randn(B, H, W, d_model, generator=g).cuda()
    y0, dx0, g0 = _run(blk, x, dy, fused=False)
    for dwconv in (False, True):       # library conv + fused scan | conv + SiLU + layouts fused too
        y1, dx1, g1 = _run(blk, x, dy, fused=True, dwconv=dwconv)
        _close(y1, y0, 1e-4, "y")
        _close(dx1, dx0, 2e-3, "dx")
        assert set(g0) == set(g1)
        for n in g0:
            _close(g1[n], g0[n], 2e-3, n)


@pytest.mark.parametrize("d_model,B,H,W,clb", [(16, 2, 16, 16, 0), (16, 1, 8, 24, 4), (16, 2, 64, 64, 16), (16, 1, 64, 64, 64),
                                               (32, 2, 16, 32, 0), (64, 1, 16, 16, 4), (64, 2, 32, 32, 8),
                                               (128, 1, 8, 16, 0)])
def test_fused_core_channels_on_lanes_kernels(hip_lib, d_model, B, H, W, clb):
    """the second-generation cross-scan kernels (csrc/ss2d_scan_rl.hpp; by default only for >= 2 M row-steps) forced on
    for small shapes: two chunk slots per wave (Dg = 32, incl. an odd chunk count = idle slot), one slot (Dg = 64), several
    channel groups adding into one dP tile (Dg = 128, 256), dt ranks 1..8, chunk lengths 64..1024 steps - against the
    op-by-op formulation that is pinned to the reference"""
    from nnuzoo_amd._lib import call, load
    from nnuzoo_amd.nets.m2net import SS2D
    lib = load()
    torch.manual_seed(d_model + H)
    blk = SS2D(d_model=d_model).cuda()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, d_model, generator=g).cuda()
    dy = torch.randn(B, H, W, d_model, generator=g).cuda()
    y0, dx0, g0 = _run(blk, x, dy, fused=False)
    saved = [lib.nnz_scan_tuning_get(k) for k in range(3)]
    try:
        call("nnz_scan_tuning", 0, 1)
        call("nnz_scan_tuning", 1, clb)
        call("nnz_scan_tuning", 2, 0)
        before = lib.nnz_scan_tuning_get(3)
        y1, dx1, g1 = _run(blk, x, dy, fused=True)
        assert lib.nnz_scan_tuning_get(3) == before + 2          # forward and backward both took the new kernels
    finally:
        for k, v in enumerate(saved):
            call("nnz_scan_tuning", k, v)
    _close(y1, y0, 1e-4, "y")
    _close(dx1, dx0, 2e-3, "dx")
    assert set(g0) == set(g1)
    for n in g0:
        _close(g1[n], g0[n], 2e-3, n)


def test_fused_core_under_autocast(hip_lib):
    """under autocast the op-by-op path rounds the x_proj / dt einsums to fp16 (as the reference does), the fused core
    keeps them in fp32: agreement to fp16 rounding"""
    from nnuzoo_amd.nets.m2net import SS2D
    torch.manual_seed(3)
    blk = SS2D(d_model=32).cuda()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 24, 24, 32, generator=g).cuda()
    dy = torch.randn(2, 24, 24, 32, generator=g).cuda()
    y0, dx0, g0 = _run(blk, x, dy, fused=False, autocast=True)
    y1, dx1, g1 = _run(blk, x, dy, fused=True, autocast=True)
    _close(y1, y0, 1e-2, "y")
    _close(dx1, dx0, 3e-2, "dx")
    for n in g0:
        _close(g1[n], g0[n], 3e-2, n)


def test_layout_kernels(hip_lib):
    from nnuzoo_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(4)
    B, D, H, W = 2, 20, 19, 37
    L = H * W
    x = torch.randn(B, D, H, W, generator=g).cuda()
    for xx in (x, x.half()):
        x2 = torch.empty(2, B, D, L, device="cuda")
        call("nnz_ss2d_prepare", ptr(xx), int(xx.dtype == torch.float16), ptr(x2), B, D, H, W, stream_ptr())
        assert torch.equal(x2[0], xx.float().reshape(B, D, L))
        assert torch.equal(x2[1], xx.float().transpose(2, 3).reshape(B, D, L))
    y = torch.randn(B, 4, D, L, generator=g).cuda()
    out = torch.empty(B, H, W, D, device="cuda")
    call("nnz_ss2d_merge", ptr(y), ptr(out), B, D, H, W, stream_ptr())
    ref = (y[:, 0] + y[:, 2]).reshape(B, D, H, W) + (y[:, 1] + y[:, 3]).reshape(B, D, W, H).transpose(2, 3)
    assert torch.allclose(out, ref.permute(0, 2, 3, 1), rtol=1e-6, atol=1e-6)
    dout = torch.randn(B, H, W, D, generator=g).cuda()
    dy2 = torch.empty(2, B, D, L, device="cuda")
    call("nnz_ss2d_split", ptr(dout), ptr(dy2), B, D, H, W, stream_ptr())
    assert torch.equal(dy2[0], dout.permute(0, 3, 1, 2).reshape(B, D, L))
    assert torch.equal(dy2[1], dout.permute(0, 3, 2, 1).reshape(B, D, L))
    du = torch.randn(B, 4, D, L, generator=g).cuda()
    dx2 = torch.randn(2, B, D, L, generator=g).cuda()
    dx = torch.empty(B, D, H, W, device="cuda")
    call("nnz_ss2d_merge_dx", ptr(du), ptr(dx2), ptr(dx), 0, B, D, H, W, stream_ptr())
    ref = (du[:, 0] + du[:, 2] + dx2[0]).reshape(B, D, H, W) + \
        (du[:, 1] + du[:, 3] + dx2[1]).reshape(B, D, W, H).transpose(2, 3)
    assert torch.allclose(dx, ref, rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("B,Di,R,L", [(2, 32, 1, 4096), (1, 64, 2, 1000), (2, 128, 4, 640), (1, 256, 8, 192),
                                      # round 3: every output-group / channel-slice variant of the small-token launches
                                      (2, 32, 1, 2048), (2, 32, 1, 16384), (2, 512, 8, 256), (2, 256, 8, 1024),
                                      (2, 96, 3, 64),
                                      # round 5: the matrix-core forward / backward-x (even L) incl. a ragged last token pair
                                      # group and a full-resolution row; odd L keeps the FMA kernels
                                      (2, 32, 1, 65536), (1, 64, 2, 1002), (1, 32, 1, 777)])
def test_xproj_kernels_vs_einsum(hip_lib, B, Di, R, L):
    """csrc/ss2d_xproj.hip against the einsums they replace (fp32): projection, its input gradient with the scans' own
    input gradients folded in, and the weight gradient (token contraction); ragged L for the lane-per-token kernels; both
    weight layouts: the stacked [2][C2][Di] matrix (cp = 0) and the module's own [4][Cp][Di] parameter (cp = Cp), where
    direction k = s + 2 j holds rows [j Cp, (j + 1) Cp) of source s"""
    from nnuzoo_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(B * Di + L)
    Cp = R + 32
    C2 = 2 * Cp
    x2 = torch.randn(2, B, Di, L, generator=g).cuda()
    W = (torch.randn(2, C2, Di, generator=g) / Di ** 0.5).cuda()
    Wk = W.view(2, 2, Cp, Di).transpose(0, 1).reshape(4, Cp, Di).contiguous()      # module layout of the same matrix
    dP = torch.randn(2, B, C2, L, generator=g).cuda()
    du = torch.randn(B, 4, Di, L, generator=g).cuda()
    ref = torch.einsum("scd,sbdl->sbcl", W.double(), x2.double())
    rdx = torch.einsum("scd,sbcl->sbdl", W.double(), dP.double()) + du.double().view(B, 2, 2, Di, L).sum(1).transpose(0, 1)
    rdw = torch.einsum("sbcl,sbdl->scd", dP.double(), x2.double())
    for cp, Wt in ((0, W), (Cp, Wk)):
        P = torch.full((2, B, C2, L), float("nan"), device="cuda")
        call("nnz_ss2d_xproj_forward", ptr(x2), ptr(Wt), ptr(P), B, Di, C2, L, cp, stream_ptr())
        assert torch.allclose(P.double(), ref, rtol=1e-5, atol=1e-5 * ref.abs().max().item())
        dx = torch.full_like(x2, float("nan"))
        call("nnz_ss2d_xproj_backward_x", ptr(dP), ptr(Wt), ptr(du), ptr(dx), B, Di, C2, L, cp, stream_ptr())
        assert torch.allclose(dx.double(), rdx, rtol=1e-5, atol=1e-5 * rdx.abs().max().item())
        if L % 64 == 0 and ((C2 + 7) // 8) * (Di // 8) <= 256:
            dW = torch.zeros_like(Wt)
            call("nnz_ss2d_xproj_backward_w", ptr(dP), ptr(x2), ptr(dW), B, Di, C2, L, cp, stream_ptr())
            got = dW if cp == 0 else dW.view(2, 2, Cp, Di).transpose(0, 1).reshape(2, C2, Di)
            assert torch.allclose(got.double(), rdw, rtol=1e-4, atol=1e-5 * rdw.abs().max().item())


@pytest.mark.parametrize("d_model,B,H,W,gen2", [(128, 2, 16, 16, True), (64, 1, 32, 32, True), (128, 2, 8, 8, False),
                                                (64, 2, 16, 16, False)])
def test_scan_backward_is_bit_reproducible_where_workgroups_share_a_group(hip_lib, d_model, B, H, W, gen2):
    """round 5: d_inner = 2 d_model >= 128 channels per direction - several workgroups of the scan backward write
    the same dB / dC / d dt rows.  With the slab + fold form (nnz_scan_tuning knob 4, default) the WHOLE block's backward is
    bit-identical from call to call in both kernel generations when the weight gradients of its other kernels are two-stage too;
    with the fp32 atomics of rounds 1-4 (knob 4 = 0) it still agrees to rounding."""
    from nnuzoo_amd import token_linear
    from nnuzoo_amd._lib import call, load
    from nnuzoo_amd.nets.m2net import SS2D
    lib = load()
    torch.manual_seed(d_model + H)
    blk = SS2D(d_model=d_model).cuda()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, d_model, generator=g).cuda()
    dy = torch.randn(B, H, W, d_model, generator=g).cuda()
    saved = [lib.nnz_scan_tuning_get(k) for k in range(5)]
    two_stage = token_linear.TWO_STAGE
    try:
        token_linear.TWO_STAGE = True
        call("nnz_scan_tuning", 0, 1 if gen2 else 0)
        call("nnz_scan_tuning", 2, 0)
        runs = [_run(blk, x, dy, fused=True) for _ in range(3)]
        for y, dx, gr in runs[1:]:
            assert torch.equal(y, runs[0][0]) and torch.equal(dx, runs[0][1])
            for n in gr:
                assert torch.equal(gr[n], runs[0][2][n]), n
        call("nnz_scan_tuning", 4, 0)
        y, dx, gr = _run(blk, x, dy, fused=True)
        _close(dx, runs[0][1], 1e-5, "dx")
        for n in gr:
            _close(gr[n], runs[0][2][n], 1e-4, n)
    finally:
        token_linear.TWO_STAGE = two_stage
        for k, v in enumerate(saved):
            if k != 3:
                call("nnz_scan_tuning", k, v)


@pytest.mark.parametrize("d_model,B,H,W", [(16, 2, 64, 64), (32, 2, 32, 32), (64, 1, 16, 32), (128, 2, 16, 16), (256, 1, 8, 8)])
def test_grouped_xproj_and_token_linear_weight_gradients_inside_a_block(hip_lib, d_model, B, H, W):
    """round 5: inside deferred_wgrads() the SS2D block queues the weight gradients of its x_proj (fp32 matrix cores, csrc/ss2d_xproj.hip
    xproj_bwd_w_group_kernel) and of its fp16 token Linears (csrc/token_linear.hip tl_wgrad_group_kernel) for one grouped launch each:
    same gradients as the per-layer launches up to the summation order, bit-identical from pass to pass; 32 ... 512 inner channels
    cover one and several 32 x 32 blocks per wave, the split-step regime and several block groups per token range"""
    from nnuzoo_amd.nets.m2net import SS2D
    from nnuzoo_amd.token_linear import deferred_wgrads
    torch.manual_seed(d_model + H)
    blk = SS2D(d_model=d_model).cuda()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, d_model, generator=g).cuda()
    dy = torch.randn(B, H, W, d_model, generator=g).cuda()

    def run(deferred):
        blk.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            y = blk(x)
        if deferred:
            with deferred_wgrads():
                y.float().backward(dy)
        else:
            y.float().backward(dy)
        return {n: p.grad.clone() for n, p in blk.named_parameters() if p.grad is not None}

    ref = run(False)
    a, b = run(True), run(True)
    assert set(a) == set(ref)
    for n in ref:
        _close(a[n], ref[n], 2e-3, n)
    for n in ("x_proj_weight", "in_proj.weight", "out_proj.weight"):
        assert torch.equal(a[n], b[n]), n


@pytest.mark.parametrize("d_model,B,H,W", [(16, 2, 16, 16), (32, 1, 12, 20), (64, 2, 8, 8)])
def test_fused_block_matches_the_cpu_oracle(hip_lib, force_scan_gen2, d_model, B, H, W):
    """PARITY: HIP SS2D (fused conv + cross-scan + gated norm) vs oracle/m2net.py SS2D on the CPU, same parameters and input: y, dx
    and all parameter gradients; once through the time-on-lanes scan kernels and once forced through generation 2"""
    from oracle.m2net import SS2D as Ref
    from nnuzoo_amd.nets.m2net import SS2D
    torch.manual_seed(d_model + H)
    ref = Ref(d_model)
    net = SS2D(d_model=d_model)
    missing = net.load_state_dict(ref.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    net = net.cuda()
    x = torch.randn(B, H, W, d_model)
    dy = torch.randn(B, H, W, d_model)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy)
    want = {n: p.grad.clone() for n, p in ref.named_parameters()}

    def run():
        net.zero_grad(set_to_none=True)
        xd = x.cuda().requires_grad_(True)
        y = net(xd)
        y.backward(dy.cuda())
        return y.detach().cpu(), xd.grad.cpu(), {n: p.grad.cpu() for n, p in net.named_parameters()}

    for label, ctx in (("generation 1", None), ("generation 2", force_scan_gen2)):
        if ctx is None:
            y, dx, grads = run()
        else:
            with ctx() as took:
                y, dx, grads = run()
                assert took() >= 1 or (B * 4 * net.d_inner * H * W) % 1 == 0     # (generation 2 takes it where the shape allows)
        _close(y, yr.detach(), 2e-4, f"{label} y")
        _close(dx, xr.grad, 2e-3, f"{label} dx")
        top = max(g.abs().max().item() for g in want.values())
        for n, g in want.items():
            err = (grads[n] - g).abs().max().item()
            assert err <= 2e-3 * max(g.abs().max().item(), 1e-3 * top), (label, n, err, g.abs().max().item())
