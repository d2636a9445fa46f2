"""FusedAdamW (training/fused_adamw.py + csrc/optimizer.hip nnz_adamw_fused): GradScaler.unscale_ + clip_grad_norm_(12) + AdamW
step of the X^2-Net plugins (reference nnUNetTrainerM2Net.py:58-65 behind nnUNetTrainer.py:1131-1139) as two launches over a
device chunk table - against torch's own sequence on the same gradients."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(7,), (33, 5), (16, 16, 3, 3), (20000,), (3, 1), (129, 65)]     # one tensor larger than a chunk, odd sizes
    return [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes] + \
        [torch.nn.Parameter(torch.randn(5, generator=g).cuda())]              # the last one never receives a gradient


def _grads(ps, step, scale, poison=False):
    g = torch.Generator().manual_seed(100 + step)
    for i, p in enumerate(ps[:-1]):
        p.grad = (torch.randn(p.shape, generator=g) * (3.0 if i == 3 else 0.5)).cuda() * scale
    if poison:
        ps[1].grad[2, 3] = float("inf")


@pytest.mark.parametrize("use_scaler", [True, False])
def test_fused_tail_equals_torch_sequence(hip_lib, use_scaler):
    from nnuzoo_amd.training.fused_adamw import FusedAdamW
    pa, pb = _params(1), _params(1)
    kw = dict(lr=3e-3, weight_decay=5e-2, eps=1e-5, betas=(0.9, 0.999))
    ref = torch.optim.AdamW(pa, **kw)
    opt = FusedAdamW(pb, **kw)
    assert opt.fused_available()
    scale = torch.full((1,), 1024.0, device="cuda")
    skipped = []
    for step in range(6):
        poison = use_scaler and step == 2
        s = float(scale) if use_scaler else 1.0
        _grads(pa, step, s, poison)
        _grads(pb, step, s, poison)
        # torch: unscale -> inf check -> clip -> step
        inv = 1.0 / s
        bad = any(not torch.isfinite(p.grad).all() for p in pa[:-1])
        for p in pa[:-1]:
            p.grad.mul_(inv)
        if not bad:
            torch.nn.utils.clip_grad_norm_(pa, 12)
            ref.step()
        found = opt.fused_step(scale.reciprocal() if use_scaler else None, 12)
        skipped.append(float(found) > 0)
        assert skipped[-1] == bad, (step, float(found))
    assert skipped == [False, False, use_scaler, False, False, False]
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=2e-6, atol=4e-7), (a - b).abs().max().item()
    assert torch.equal(pb[-1], _params(1)[-1])                                   # no gradient: untouched (no decay either)
    for a, b in zip(pa[:-1], pb[:-1]):
        sa, sb = ref.state[a], opt.state[b]
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=2e-6, atol=1e-8)
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=2e-6, atol=1e-10)
        assert float(sb["step"]) == float(sa["step"]) == (5.0 if use_scaler else 6.0)
    assert 1 <= opt.table_builds <= 6            # eager gradients are new tensors: the table follows whenever an address moved


def test_state_dict_round_trip_and_static_gradients(hip_lib):
    """checkpoint compatibility (torch's state names) and the replay situation: gradients rewritten in place keep the table"""
    from nnuzoo_amd.training.fused_adamw import FusedAdamW
    ps = _params(3)
    opt = FusedAdamW(ps, lr=1e-3, weight_decay=5e-2, eps=1e-5)
    _grads(ps, 0, 1.0)
    opt.fused_step(None, 12)
    static = [p.grad for p in ps[:-1]]
    for k in range(1, 4):
        g = torch.Generator().manual_seed(50 + k)
        for t in static:
            t.copy_(torch.randn(t.shape, generator=g))
        opt.fused_step(None, 12)
    assert opt.table_builds == 1
    sd = copy.deepcopy(opt.state_dict())
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 4.0
    before = [p.detach().clone() for p in ps]
    opt2 = FusedAdamW(ps, lr=1e-3, weight_decay=5e-2, eps=1e-5)
    opt2.load_state_dict(sd)
    ref = torch.optim.AdamW([torch.nn.Parameter(b.clone()) for b in before], lr=1e-3, weight_decay=5e-2, eps=1e-5)
    ref.load_state_dict(copy.deepcopy(sd))
    for q, p in zip(ref.param_groups[0]["params"], ps):
        q.grad = None if p.grad is None else p.grad.clone()
    torch.nn.utils.clip_grad_norm_(ref.param_groups[0]["params"], 12)
    ref.step()
    opt2.fused_step(None, 12)
    for q, p in zip(ref.param_groups[0]["params"], ps):
        assert torch.allclose(q, p, rtol=2e-6, atol=2e-7)
    assert float(opt2.state[ps[0]]["step"]) == 5.0


def test_freeze_then_unfreeze_keeps_every_parameters_own_step_counter(hip_lib):
    """the bias corrections come from the chunk owner's OWN step counter.  Three steps with parameters 0-2 frozen (no
    gradient, the Swin-UMamba plugins' first epochs), then three with everything trainable: the newly unfrozen parameters are at
    step 1 while the others are at step 4 - against torch.optim.AdamW, which keeps one counter per parameter."""
    from nnuzoo_amd.training.fused_adamw import FusedAdamW
    pa, pb = _params(5), _params(5)
    kw = dict(lr=3e-3, weight_decay=5e-2, eps=1e-5, betas=(0.9, 0.999))
    ref, opt = torch.optim.AdamW(pa, **kw), FusedAdamW(pb, **kw)
    for step in range(6):
        for ps in (pa, pb):
            _grads(ps, step, 1.0)
            if step < 3:
                for p in ps[:3]:
                    p.grad = None
        torch.nn.utils.clip_grad_norm_(pa, 12)
        ref.step()
        opt._token = None                           # what the trainers do when the trainable set changes
        assert float(opt.fused_step(None, 12)) == 0
    for i, (a, b) in enumerate(zip(pa[:-1], pb[:-1])):
        assert float(opt.state[b]["step"]) == float(ref.state[a]["step"]) == (3.0 if i < 3 else 6.0)
        assert torch.allclose(a, b, rtol=2e-6, atol=4e-7), (i, (a - b).abs().max().item())
        assert torch.allclose(ref.state[a]["exp_avg_sq"], opt.state[b]["exp_avg_sq"], rtol=2e-6, atol=1e-10)


def test_large_scaled_gradients_that_are_finite_are_not_skipped(hip_lib):
    """gradient norm 3e4 under a loss scale of 65536 - every fp32 value finite, the sum of the SCALED squares (4e18) far
    beyond the fixed-point accumulator's 2^52; torch's unscale_ / clip / step applies that step, so must the fused tail"""
    from nnuzoo_amd.training.fused_adamw import FusedAdamW
    p = torch.nn.Parameter(torch.zeros(1 << 20, device="cuda"))
    q = torch.nn.Parameter(torch.zeros(1 << 20, device="cuda"))
    g = torch.randn(1 << 20, generator=torch.Generator().manual_seed(3)).cuda() * (3e4 / 1024)     # norm ~3e4
    scale = torch.full((1,), 65536.0, device="cuda")
    p.grad = g * scale
    q.grad = g.clone()
    kw = dict(lr=1e-3, weight_decay=0.0, eps=1e-5)
    opt, ref = FusedAdamW([p], **kw), torch.optim.AdamW([q], **kw)
    assert float(opt.fused_step(scale.reciprocal(), 12)) == 0
    torch.nn.utils.clip_grad_norm_([q], 12)
    ref.step()
    assert torch.allclose(p, q, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("norm", [2.6e9, 3.0e13])
def test_huge_finite_gradient_norm_is_clipped_and_applied_like_torch(hip_lib, norm):
    """round 6: SSND2NetP in an fp32 step has a gradient norm of 2.6e9 (tools/probes/ssnd2net_fp32_probe.py) - every value finite,
    the sum of squares (7e18) beyond the 2^52 of the fine fixed-point record.  torch's clip_grad_norm_(12) scales such a gradient
    to norm 12 and steps; the fused tail skipped the step (24 of 24).  The wide-range record makes it clip and step."""
    from nnuzoo_amd.training.fused_adamw import FusedAdamW
    n = 1 << 20
    p = torch.nn.Parameter(torch.zeros(n, device="cuda"))
    q = torch.nn.Parameter(torch.zeros(n, device="cuda"))
    g = torch.randn(n, generator=torch.Generator().manual_seed(4)).cuda() * (norm / 1024)
    p.grad, q.grad = g.clone(), g.clone()
    kw = dict(lr=1e-3, weight_decay=0.0, eps=1e-5)
    opt, ref = FusedAdamW([p], **kw), torch.optim.AdamW([q], **kw)
    found = [float(opt.fused_step(None, 12)) for _ in range(2)]
    assert found == [0.0, 0.0]
    for _ in range(2):
        torch.nn.utils.clip_grad_norm_([q], 12)
        ref.step()
        q.grad = g.clone()
    assert float(q.abs().max()) > 0
    assert torch.allclose(p, q, rtol=2e-5, atol=1e-9), (p - q).abs().max().item()
