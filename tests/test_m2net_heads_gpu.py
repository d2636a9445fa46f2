"""The convolutions of an fp16-autocast X^2-Net step that ran on MIOpen until round 6, on this package's kernels:

* 3x3 side heads (`side1 .. side6`, /root/reference/nnunetv2/nets/m2net.py:874-880) as one 32-channel block of the tap-table conv
  kernels (nnuzoo_amd/rebnconv.py `_Head3x3Fn`),
* the fuse convolution (`outconv`, m2net.py:881, 948-950) on csrc/sepconv32.hip head1x1_* with fp16 activations,
* a REBNCONV on its own (the `rebnconvin` of every MU stage, m2net.py:18-30) incl. the 1-channel network input zero-padded to 32 channels,
* 1x1 patch embeddings / stage outputs on fp16 token rows (csrc/dense32.hip *_h16).

Reference of every case: the same torch modules in fp32 on the fp16-rounded operands (what autocast feeds its kernels), i.e. the plain
PyTorch fp32 formulation of the op; tolerance = fp16 output rounding (2^-10 of the tensor's scale, x 4)."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

pytestmark = pytest.mark.gpu

HALF_EPS = 2.0 ** -10


def _close(got, ref, what, k=4.0, floor=0.0):
    scale = max(ref.abs().max().item(), floor, 1e-30)
    err = (got.float() - ref.float()).abs().max().item()
    assert err <= k * HALF_EPS * scale, (what, err, scale)


def _close_l2(got, ref, what, tol):
    """relative L2 error: for tensors behind a ReLU whose mask fp16 rounding of the conv output flips at isolated positions (a flipped
    position moves its neighbourhood's input gradient by O(1) - the torch fp16-autocast modules differ from fp32 in the same way)"""
    err = (got.float() - ref.float()).norm().item() / max(ref.float().norm().item(), 1e-30)
    assert err <= tol, (what, err)


def _r16(t):
    return t.half().float()


@pytest.mark.parametrize("cin,cout,hw,layout", [(32, 2, (40, 48), "nchw"), (64, 2, (64, 64), "tokens"), (128, 5, (24, 40), "tokens"),
                                                (512, 3, (16, 16), "nchw")])
def test_side_head_3x3_matches_the_fp32_convolution(hip_lib, cin, cout, hw, layout):
    from nnuzoo_amd import rebnconv
    torch.manual_seed(cin + cout)
    conv = nn.Conv2d(cin, cout, 3, padding=1).cuda()
    H, W = hw
    if layout == "tokens":          # what the stages hand over: an NCHW view of token-major fp16 storage
        x = torch.randn(2, H, W, cin, device="cuda").half().permute(0, 3, 1, 2)
    else:
        x = torch.randn(2, cin, H, W, device="cuda").half()
    x.requires_grad_(True)
    g = torch.randn(2, cout, H, W, device="cuda").half()
    with torch.autocast("cuda", dtype=torch.float16):
        assert rebnconv.head3x3_ok(conv, x)
        y = rebnconv.head3x3(conv, x)
    assert y.dtype == torch.float16 and y.shape == (2, cout, H, W) and y.is_contiguous()
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    wr = _r16(conv.weight.detach()).requires_grad_(True)
    br = conv.bias.detach().clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=1)
    yr.backward(g.float())
    _close(y, yr, "y")
    _close(x.grad, xr.grad, "dx")
    # weight gradient: a sum over 2 H W positions of fp16 products accumulated in fp32 - the error is the operands' rounding only
    _close(conv.weight.grad, wr.grad, "dw", k=1.0, floor=1e-3)
    _close(conv.bias.grad, br.grad, "db", k=1.0, floor=1e-3)


def test_side_head_3x3_repeats_bit_for_bit(hip_lib):
    from nnuzoo_amd import rebnconv
    torch.manual_seed(0)
    conv = nn.Conv2d(64, 2, 3, padding=1).cuda()
    x = torch.randn(2, 96, 96, 64, device="cuda").half().permute(0, 3, 1, 2).requires_grad_(True)
    g = torch.randn(2, 2, 96, 96, device="cuda").half()
    res = []
    for _ in range(2):
        conv.zero_grad(set_to_none=True)
        x.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            y = rebnconv.head3x3(conv, x)
        y.backward(g)
        res.append((y.detach().clone(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("K,N,hw", [(12, 2, (64, 80)), (18, 3, (33, 47)), (48, 8, (16, 16))])
def test_fuse_convolution_1x1_with_fp16_activations(hip_lib, K, N, hw):
    from nnuzoo_amd import sepconv32
    torch.manual_seed(K)
    conv = nn.Conv2d(K, N, 1).cuda()
    H, W = hw
    x = torch.randn(2, K, H, W, device="cuda").half().requires_grad_(True)       # the concatenated side outputs: plain NCHW
    g = torch.randn(2, N, H, W, device="cuda").half()
    with torch.autocast("cuda", dtype=torch.float16):
        assert sepconv32.head1x1_ok(conv, x)
        y = sepconv32.head1x1(conv, x)
    assert y.dtype == torch.float16
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    wr = conv.weight.detach().clone().requires_grad_(True)        # the kernels read the fp32 master weight itself
    br = conv.bias.detach().clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br)
    yr.backward(g.float())
    _close(y, yr, "y")
    _close(x.grad, xr.grad, "dx")
    _close(conv.weight.grad, wr.grad, "dw", k=0.5, floor=1e-3)
    _close(conv.bias.grad, br.grad, "db", k=0.5, floor=1e-3)


@pytest.mark.parametrize("cin,cout,hw,dil", [(1, 32, (64, 64), 1), (3, 64, (40, 56), 1), (64, 64, (32, 48), 1), (128, 64, (24, 24), 2)])
def test_rebnconv_on_its_own_matches_the_torch_modules(hip_lib, cin, cout, hw, dil):
    """training mode: batch statistics, the running estimates, every gradient; fp32 torch modules on fp16-rounded operands as reference"""
    import copy
    from nnuzoo_amd import rebnconv
    from nnuzoo_amd.nets.common2d import REBNCONV
    torch.manual_seed(cin * 7 + cout)
    mod = REBNCONV(cin, cout, dirate=dil).cuda().train()
    with torch.no_grad():
        mod.bn_s1.weight.uniform_(0.5, 1.5)
        mod.bn_s1.bias.uniform_(-0.3, 0.3)
    ref = copy.deepcopy(mod)
    with torch.no_grad():
        ref.conv_s1.weight.copy_(_r16(ref.conv_s1.weight))
    H, W = hw
    x = torch.randn(2, cin, H, W, device="cuda")
    x16 = _r16(x)
    xa = x16.clone().requires_grad_(True)
    xb = x16.clone().requires_grad_(True)
    g = torch.randn(2, cout, H, W, device="cuda").half()
    with torch.autocast("cuda", dtype=torch.float16):
        assert rebnconv.unit_ok(mod, xa)
        y = mod(xa)
    assert mod.backend == "hip" and y.dtype == torch.float16
    assert y.permute(0, 2, 3, 1).is_contiguous()            # NCHW view of channels-last storage: the patch embedding reads it as tokens
    y.backward(g)
    yr = ref(xb)
    yr.backward(g.float())
    _close(y, yr, "y", k=8.0)           # the raw conv output is rounded to fp16 before the normalisation, like autocast's
    _close_l2(xa.grad, xb.grad, "dx", 1e-2)
    _close_l2(mod.conv_s1.weight.grad, ref.conv_s1.weight.grad, "dw", 1e-2)
    _close_l2(mod.bn_s1.weight.grad, ref.bn_s1.weight.grad, "dgamma", 1e-2)
    _close_l2(mod.bn_s1.bias.grad, ref.bn_s1.bias.grad, "dbeta", 1e-2)
    assert torch.allclose(mod.bn_s1.running_mean, ref.bn_s1.running_mean, atol=2e-3)
    assert torch.allclose(mod.bn_s1.running_var, ref.bn_s1.running_var, rtol=5e-3, atol=1e-4)
    assert int(mod.bn_s1.num_batches_tracked) == int(ref.bn_s1.num_batches_tracked) == 1


def test_rebnconv_on_its_own_in_eval_mode_uses_the_running_estimates(hip_lib):
    import copy
    from nnuzoo_amd.nets.common2d import REBNCONV
    torch.manual_seed(3)
    mod = REBNCONV(64, 32).cuda().eval()
    with torch.no_grad():
        mod.bn_s1.running_mean.uniform_(-0.2, 0.2)
        mod.bn_s1.running_var.uniform_(0.5, 1.5)
    ref = copy.deepcopy(mod)
    with torch.no_grad():
        ref.conv_s1.weight.copy_(_r16(ref.conv_s1.weight))
    x = _r16(torch.randn(2, 64, 40, 40, device="cuda"))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        y = mod(x)
    assert mod.backend == "hip"
    _close(y, ref(x), "y", k=8.0)


@pytest.mark.parametrize("K,N,tokens", [(64, 16, (2, 64, 64)), (32, 16, (2, 128, 96)), (16, 128, (2, 24, 24)), (128, 256, (2, 8, 8))])
def test_pointwise_convolution_on_fp16_token_rows(hip_lib, K, N, tokens):
    from nnuzoo_amd import sepconv32
    torch.manual_seed(K + N)
    conv = nn.Conv2d(K, N, 1).cuda()
    B, H, W = tokens
    x = torch.randn(B, H, W, K, device="cuda").half().requires_grad_(True)
    g = torch.randn(B, H, W, N, device="cuda").half()
    with torch.autocast("cuda", dtype=torch.float16):
        assert sepconv32.pointwise_ok(conv, x)
        y = sepconv32.pointwise_tokens(conv, x)
    assert y.dtype == torch.float16 and y.shape == (B, H, W, N)
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    wr = conv.weight.detach().clone().requires_grad_(True)
    br = conv.bias.detach().clone().requires_grad_(True)
    yr = F.conv2d(xr.permute(0, 3, 1, 2), wr, br).permute(0, 2, 3, 1)
    yr.backward(g.float())
    _close(y, yr, "y")
    _close(x.grad, xr.grad, "dx")
    _close(conv.weight.grad, wr.grad, "dw", k=0.5, floor=1e-3)
    _close(conv.bias.grad, br.grad, "db", k=0.5, floor=1e-3)


@pytest.mark.parametrize("trainer,size", [("nnUNetTrainerM2NetP", 128), ("nnUNetTrainerM2Net", 256)])
def test_m2net_steps_call_no_torch_convolution_or_batchnorm_module(hip_lib, trainer, size):
    """forward hooks on EVERY nn.Conv2d / nn.BatchNorm2d of the net: a training step must not run the forward of any of them - MU
    stems, 1x1 patch embeddings and stage outputs, RSU4F units, depthwise convolutions of the SS2D blocks, side heads and the fuse
    convolution all go through this package's kernels, which read the modules' parameters; and the parameter shadow leaves the
    parameters of exactly those modules fp32"""
    from nnuzoo_amd.param_shadow import _eligible
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training import zoo_trainers
    plans, cfg, dj = nnunet_plans(2, (size, size), batch_size=2)
    torch.manual_seed(0)
    tr = getattr(zoo_trainers, trainer)(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    net = tr.network.module if hasattr(tr.network, "module") else tr.network
    ran = []
    for n, m in net.named_modules():
        if isinstance(m, (nn.Conv2d, nn.BatchNorm2d)):
            m.register_forward_hook(lambda mod, a, o, n=n: ran.append(n))
    b = synthetic_batch(2, (size, size), tr._get_deep_supervision_scales(), seed=3)
    b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
    out = tr.train_step(b)
    torch.cuda.synchronize()
    assert torch.isfinite(torch.as_tensor(out["loss"])).all()
    # M2NetP's three RSU4F stages have 16-channel units: the tap-table conv kernels work on 32-channel blocks, those stay on the library
    allowed = ("stage5.", "stage6.", "stage5d.") if trainer.endswith("P") else ()
    left = sorted({n for n in ran if not n.startswith(allowed)}) if allowed else sorted(set(ran))
    assert not left, left
    names, _ = _eligible(net)
    assert not names, names          # nothing left for the fp16 parameter shadow in these nets
