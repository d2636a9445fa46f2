"""The fp32 depthwise-separable conv -> BatchNorm -> ReLU unit of SwT2Net's RSU4F stages on hand-written kernels
(nnuzoo_amd/sepconv32.py, csrc/sepconv32.hip + the fp32 MFMA Linear kernels) against the torch modules the reference's classes are
made of (/root/reference/nnunetv2/nets/swt2net.py:17-31 REBNCONV, :873-905 RSU4F) in float64 on the CPU.  The whole-net goldens of
tests/test_zoo_gpu.py (SwT2Net forward / backward from the reference's own class) run through the same path."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(got, ref, tol, what):
    ref = ref.to(torch.float64)
    err = (got.detach().double().cpu() - ref).abs().max().item()
    assert err <= tol * max(ref.abs().max().item(), 1e-12), (what, err, ref.abs().max().item())


@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 512), (1, 8, 8, 1024), (2, 9, 13, 36), (1, 1, 5, 4), (2, 64, 64, 64), (2, 256, 256, 32)])
def test_depthwise_3x3_forward_backward(hip_lib, B, H, W, C):
    from nnuzoo_amd.sepconv32 import _Dw3x3Fn
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, H, W, C, generator=g)
    w = torch.randn(C, 1, 3, 3, generator=g) * 0.3
    b = torch.randn(C, generator=g)
    dy = torch.randn(B, H, W, C, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=1, groups=C).permute(0, 2, 3, 1)
    rx, rw, rb = torch.autograd.grad(ref, [xr, wr, br], dy.double())
    runs = []
    for _ in range(2):
        xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
        y = _Dw3x3Fn.apply(xd, wd, bd)
        runs.append((y.detach(),) + torch.autograd.grad(y, [xd, wd, bd], dy.to(DEV)))
    for name, got, want in zip(("y", "dx", "dw", "db"), runs[0], (ref.detach(), rx, rw, rb)):
        _close(got, want, 2e-5, name)
    assert all(torch.equal(p, q) for p, q in zip(runs[0], runs[1]))          # fixed-order reductions


@pytest.mark.parametrize("T,C,training", [(512, 512, True), (128, 1024, True), (77, 36, True), (300, 40, False), (16, 4, True)])
def test_batchnorm_relu_forward_backward(hip_lib, T, C, training):
    from nnuzoo_amd.sepconv32 import _BnReluFn
    g = torch.Generator().manual_seed(T + C)
    x = torch.randn(T, C, generator=g) * 1.7 + torch.randn(C, generator=g) * 3        # means large against the spread too
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    dy = torch.randn(T, C, generator=g)
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, gamma, beta))
    rm64, rv64 = rm.double().clone(), rv.double().clone()
    ref = F.relu(F.batch_norm(xr, rm64, rv64, gr, br, training, 0.1, 1e-5))
    xd, gd, bd = (t.to(DEV).requires_grad_(True) for t in (x, gamma, beta))
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    y = _BnReluFn.apply(xd, gd, bd, rmd, rvd, 0.1, 1e-5, training)
    _close(y, ref.detach(), 2e-5, "y")
    _close(rmd, rm64, 1e-5, "running_mean")
    _close(rvd, rv64, 1e-5, "running_var")
    if training:
        want = torch.autograd.grad(ref, [xr, gr, br], dy.double())
        got = torch.autograd.grad(y, [xd, gd, bd], dy.to(DEV))
        for name, a, b in zip(("dx", "dgamma", "dbeta"), got, want):
            _close(a, b, 5e-5, name)


def _rsu(mid, cin, cout):
    from nnuzoo_amd.nets.swt2net import RSU4F
    torch.manual_seed(3)
    m = RSU4F(cin, mid, cout)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.uniform_(0.5, 1.5) if p is not None and p.mean() > 0.5 else p.normal_(0, 0.2)
    return m


@pytest.mark.parametrize("cin,mid,cout,size", [(64, 32, 64, 16), (512, 256, 512, 8)])
def test_rsu4f_matches_the_torch_modules(hip_lib, cin, mid, cout, size):
    """the whole stage: HIP path on the device vs the module path (stock torch convolutions / batch norm) in float64 on the CPU -
    output, dx, every parameter gradient, the running statistics; bit-identical run to run"""
    import copy
    from nnuzoo_amd import sepconv32
    ref = _rsu(mid, cin, cout).double().train()
    x = torch.randn(2, cin, size, size, generator=torch.Generator().manual_seed(5))
    dy = torch.randn(2, cout, size, size, generator=torch.Generator().manual_seed(6))
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy.double())
    runs = []
    for _ in range(2):
        net = _rsu(mid, cin, cout).to(DEV).train()
        xd = x.to(DEV).requires_grad_(True)
        assert sepconv32.hip_path_ok(net, xd)
        y = net(xd)
        assert net.backend == "hip-f32"
        y.backward(dy.to(DEV))
        runs.append((net, y.detach(), xd.grad))
    net, y, dx = runs[0]
    _close(y, yr.detach(), 1e-4, "y")
    _close(dx, xr.grad, 3e-4, "dx")
    top = max(p.grad.abs().max().item() for p in ref.parameters())
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        err = (p.grad.double().cpu() - q.grad).abs().max().item()
        assert err <= 3e-4 * max(q.grad.abs().max().item(), 1e-3 * top), (n, err, q.grad.abs().max().item())
    for (n, b), (_, c) in zip(net.named_buffers(), ref.named_buffers()):
        _close(b.double() if b.is_floating_point() else b.double(), c.double(), 1e-5, n)
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    assert all(torch.equal(p.grad, q.grad) for p, q in zip(runs[0][0].parameters(), runs[1][0].parameters()))


@pytest.mark.parametrize("T,K,N", [(2 * 128 * 128, 32, 32), (40000, 64, 64), (17000, 36, 8), (2 * 512 * 512, 32, 32)])
def test_pointwise_1x1_forward_backward(hip_lib, T, K, N):
    """the 1x1 convolution as a token Linear; at these sizes its weight gradient takes the small-channel streaming kernel"""
    from nnuzoo_amd import sepconv32
    g = torch.Generator().manual_seed(T % 1000 + K)
    x = torch.randn(1, 1, T, K, generator=g)
    w = torch.randn(N, K, 1, 1, generator=g) * 0.2
    dy = torch.randn(1, 1, T, N, generator=g)
    assert T >= sepconv32.SMALL_WGRAD_MIN_TOKENS
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = xr @ wr.view(N, K).t()
    rx, rw = torch.autograd.grad(ref, [xr, wr], dy.double())
    runs = []
    for _ in range(2):
        xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
        y = sepconv32._Pointwise1x1Fn.apply(xd, wd, None)
        runs.append((y.detach(),) + torch.autograd.grad(y, [xd, wd], dy.to(DEV)))
    _close(runs[0][0], ref.detach(), 2e-5, "y")
    _close(runs[0][1], rx, 2e-5, "dx")
    _close(runs[0][2], rw, 1e-4, "dw")
    assert all(torch.equal(p, q) for p, q in zip(runs[0], runs[1]))


@pytest.mark.parametrize("B,K,N,H,W,token_major,bias", [(2, 32, 2, 64, 64, True, True), (2, 512, 2, 8, 8, True, True),
                                                         (2, 12, 2, 48, 40, False, True), (1, 64, 3, 33, 17, True, False),
                                                         (2, 32, 2, 512, 512, True, True)])
def test_head_1x1_forward_backward(hip_lib, B, K, N, H, W, token_major, bias):
    """the side heads / fuse convolution (1x1 to a few channels) against F.conv2d in float64: token-major stage outputs (permuted
    views) and NCHW concatenations, y NCHW; dx comes back in the layout of x; bit-identical run to run"""
    from nnuzoo_amd import sepconv32
    g = torch.Generator().manual_seed(K + H)
    xs = torch.randn(B, H, W, K, generator=g) if token_major else torch.randn(B, K, H, W, generator=g)
    conv = torch.nn.Conv2d(K, N, 1, bias=bias)
    dy = torch.randn(B, N, H, W, generator=g)
    xr = (xs.permute(0, 3, 1, 2) if token_major else xs).double().requires_grad_(True)
    w64 = conv.weight.detach().double().requires_grad_(True)
    b64 = conv.bias.detach().double().requires_grad_(True) if bias else None
    ref = F.conv2d(xr, w64, b64)
    want = torch.autograd.grad(ref, [xr, w64] + ([b64] if bias else []), dy.double())
    conv = conv.to(DEV)
    runs = []
    for _ in range(2):
        xd_s = xs.to(DEV).requires_grad_(True)
        xd = xd_s.permute(0, 3, 1, 2) if token_major else xd_s
        assert sepconv32.head1x1_ok(conv, xd)
        y = sepconv32.head1x1(conv, xd)
        assert y.is_contiguous() and tuple(y.shape) == (B, N, H, W)
        got = torch.autograd.grad(y, [xd_s, conv.weight] + ([conv.bias] if bias else []), dy.to(DEV))
        runs.append((y.detach(),) + got)
    _close(runs[0][0], ref.detach(), 2e-5, "y")
    dx_want = want[0].permute(0, 2, 3, 1) if token_major else want[0]
    _close(runs[0][1], dx_want, 2e-5, "dx")
    _close(runs[0][2], want[1], 1e-4, "dw")
    if bias:
        _close(runs[0][3], want[2], 1e-4, "db")
    assert all(torch.equal(p, q) for p, q in zip(runs[0], runs[1]))


@pytest.mark.parametrize("cin,cout,size", [(1, 32, 64), (64, 64, 32), (3, 16, 24)])
def test_stage_stem_matches_the_torch_modules(hip_lib, cin, cout, size):
    """get_dwconv_layer (depthwise 3x3 + pointwise 1x1) of a Swin U-net stage, incl. the 1-channel stem of stage 1 (zero-padded to four
    channels): output, dx and both weight gradients against the module's own torch path in float64"""
    from nnuzoo_amd import sepconv32
    from nnuzoo_amd.nets.common2d import get_dwconv_layer
    torch.manual_seed(cin + cout)
    seq = get_dwconv_layer(2, cin, cout)
    x = torch.randn(2, cin, size, size)
    dy = torch.randn(2, cout, size, size)
    ref = copy_double(seq)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy.double())
    seq = seq.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    assert sepconv32.stem_ok(seq, xd)
    y = sepconv32.stem_forward(seq, xd)
    y.backward(dy.to(DEV))
    _close(y, yr.detach(), 2e-5, "y")
    _close(xd.grad, xr.grad, 5e-5, "dx")
    for (n, p), (_, q) in zip(seq.named_parameters(), ref.named_parameters()):
        _close(p.grad, q.grad, 2e-4, n)


def copy_double(m):
    import copy
    return copy.deepcopy(m).double()


@pytest.mark.parametrize("T,N,K", [(32768, 16, 32), (20000, 64, 64), (16384 + 77, 4, 8), (524288, 32, 16)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("bias", [False, True])
def test_small_channel_weight_gradient_kernel_with_bias_and_fp16_rows(hip_lib, T, N, K, dtype, bias):
    """csrc/sepconv32.hip pw_wgrad_small_kernel<TA, BIAS> through nnz_pw_wgrad_small: dW = dy^T x and db = sum_t dy[t] for fp32 or fp16
    token rows against the fp64 products of the same (rounded) rows; two launches give the same bits (fixed-order folds)"""
    import ctypes as C
    from nnuzoo_amd import _lib
    from nnuzoo_amd._lib import call, ptr, stream_ptr
    torch.manual_seed(T % 97 + N + K)
    dy = torch.randn(T, N, device="cuda").to(dtype)
    x = torch.randn(T, K, device="cuda").to(dtype)
    lib = _lib.load()
    outs = []
    for _ in range(2):
        ws = torch.empty(int(lib.nnz_pw_wgrad_small_workspace_floats_b(T, N, K, int(bias))), dtype=torch.float32, device="cuda")
        dw = torch.empty(N, K, device="cuda")
        db = torch.empty(N, device="cuda") if bias else None
        call("nnz_pw_wgrad_small", ptr(dy), ptr(x), int(dtype == torch.float16), ptr(ws), ptr(dw), ptr(db), T, N, K, stream_ptr())
        outs.append((dw, db))
    assert torch.equal(outs[0][0], outs[1][0]) and (not bias or torch.equal(outs[0][1], outs[1][1]))
    ref = dy.double().t() @ x.double()
    tol = 2e-5 * float(T) ** 0.5
    assert (outs[0][0].double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item() / float(T) ** 0.5)
    if bias:
        rb = dy.double().sum(0)
        assert (outs[0][1].double() - rb).abs().max().item() <= tol
