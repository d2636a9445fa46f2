"""One nnUNetTrainer.train_step on the HIP path vs the same step restated with the CPU oracle (fp32):
loss value, and the parameters after the SGD step (lr 1e-2, momentum .99 nesterov, wd 3e-5, clip 12)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.plain_conv_unet import OraclePlainConvUNet, planner_arch_kwargs
from oracle.losses import deep_supervision_loss
from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer


def test_train_step_matches_oracle(hip_lib):
    patch = (32, 32, 32)
    plans, cfg, dj = nnunet_plans(3, patch, batch_size=2)
    torch.manual_seed(0)
    tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    arch = plans["configurations"][cfg]["architecture"]["arch_kwargs"]
    assert arch["n_stages"] == 4
    ref = OraclePlainConvUNet(1, num_classes=2, **planner_arch_kwargs(3, 4, arch["features_per_stage"]))
    ref.load_state_dict({k: v.cpu() for k, v in tr.network.state_dict().items()})
    opt = torch.optim.SGD(ref.parameters(), 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    scales = tr._get_deep_supervision_scales()
    assert scales == [[1.0] * 3, [0.5] * 3, [0.25] * 3]
    batch = synthetic_batch(2, patch, scales, seed=11)
    losses, ref_losses = [], []
    for it in range(3):
        out = tr.train_step(batch)
        losses.append(float(out["loss"]))
        opt.zero_grad()
        l = deep_supervision_loss(ref(batch["data"]), batch["target"], batch_dice=False)
        l.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 12)
        opt.step()
        ref_losses.append(float(l))
    print("hip losses", losses, "oracle losses", ref_losses)
    assert abs(losses[0] - ref_losses[0]) < 5e-3 * max(1.0, abs(ref_losses[0]))
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 3e-2 * max(1.0, abs(b))
    assert losses[-1] < losses[0]
    # parameters after 3 steps stay close to the fp32 trajectory
    sd = tr.network.state_dict()
    num = sum(((sd[k].cpu() - v) ** 2).sum().item() for k, v in ref.state_dict().items())
    den = sum((v ** 2).sum().item() for v in ref.state_dict().values())
    assert (num / den) ** 0.5 < 2e-2


def test_validation_step_and_checkpoint(hip_lib, tmp_path):
    patch = (32, 32, 32)
    plans, cfg, dj = nnunet_plans(3, patch, batch_size=2)
    tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    batch = synthetic_batch(2, patch, tr._get_deep_supervision_scales(), seed=5)
    v = tr.validation_step(batch)
    assert v["tp_hard"].shape == (1,) and np.isfinite(v["loss"])
    d = nnUNetTrainer.pseudo_dice([v, v])
    assert 0.0 <= d[0] <= 1.0
    f = str(tmp_path / "checkpoint_latest.pth")
    tr.save_checkpoint(f)
    ck = torch.load(f, weights_only=False)
    assert set(ck) >= {"network_weights", "optimizer_state", "grad_scaler_state", "current_epoch", "init_args",
                       "trainer_name", "_best_ema", "inference_allowed_mirroring_axes", "logging"}
    tr2 = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr2.load_checkpoint(f)
    for (k, a), (_, b) in zip(tr.network.state_dict().items(), tr2.network.state_dict().items()):
        assert torch.equal(a, b), k


def _two_trainers(patch=(32, 32, 32)):
    plans, cfg, dj = nnunet_plans(3, patch, batch_size=2)
    torch.manual_seed(0)
    a = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
    a.initialize()
    b = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
    b.initialize()
    b.network.load_state_dict(a.network.state_dict())
    b.use_fused_optimizer = False
    return a, b


def test_fused_optimizer_matches_torch_tail(hip_lib):
    """unscale + clip(12) + Nesterov SGD + scaler update as the fused HIP tail.

    (1) exact check inside one trainer: after a fused step, momentum and parameters equal torch's update rule applied
        to the very gradients of that step (p.grad are views of the arena);
    (2) trajectory check against a second trainer running torch's GradScaler / clip_grad_norm_ / SGD.step sequence.
        Two runs of the same step differ by fp16 rounding (fp32 atomics in the statistics) and LeakyReLU's gradient
        amplifies sign flips of near-zero pre-activations to a few % of gradient norm, so (2) is a loose bound."""
    from nnuzoo_amd.training.fused_sgd import FusedSGD
    fused, plain = _two_trainers()
    assert isinstance(fused.optimizer, FusedSGD) and isinstance(plain.optimizer, torch.optim.SGD)
    batch = synthetic_batch(2, (32, 32, 32), fused._get_deep_supervision_scales(), seed=3)
    params = list(fused.network.parameters())
    before = [p.detach().clone() for p in params]
    scale0 = 65536.0
    fused.train_step(batch)
    assert float(fused.grad_scaler.get_scale()) == scale0
    lr, mom, wd = 1e-2, 0.99, 3e-5
    grads = [None if p.grad is None else p.grad.detach() / scale0 for p in params]
    assert sum(g is None for g in grads) == 2  # weight + bias of the lowest-resolution head (loss weight 0)
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads if g is not None)).item()
    clip = min(1.0, 12.0 / (total + 1e-6))
    assert abs(float(fused.optimizer.total_grad_norm()) / scale0 - total) < 1e-4 * total
    for p, p0, g in zip(params, before, grads):
        if g is None:  # untouched, no optimizer state - torch.optim.SGD's behaviour for a parameter without .grad
            assert torch.equal(p.detach(), p0) and "momentum_buffer" not in fused.optimizer.state.get(p, {})
            continue
        d = g * clip + wd * p0
        buf = fused.optimizer.state[p]["momentum_buffer"]
        assert torch.allclose(buf, d, rtol=1e-5, atol=1e-7 * (d.abs().max().item() + 1e-12))
        want = p0 - lr * (d + mom * d)
        assert torch.allclose(p.detach(), want, rtol=1e-5, atol=1e-8)
    plain.train_step(batch)
    for it in range(3):
        lf, lp = fused.train_step(batch)["loss"], plain.train_step(batch)["loss"]
        assert abs(float(lf) - float(lp)) < 1e-2 * max(1.0, abs(float(lp)))
    sf, sp = fused.network.state_dict(), plain.network.state_dict()
    num = sum(((sf[k] - sp[k]).float() ** 2).sum().item() for k in sp)
    den = sum((sp[k].float() ** 2).sum().item() for k in sp)
    assert (num / den) ** 0.5 < 5e-3
    stf, stp = fused.optimizer.state_dict(), plain.optimizer.state_dict()
    assert stf["state"].keys() == stp["state"].keys()
    assert all(stf["state"][k]["momentum_buffer"].shape == stp["state"][k]["momentum_buffer"].shape for k in stp["state"])
    assert float(fused.grad_scaler.get_scale()) == float(plain.grad_scaler.get_scale())


def test_fused_optimizer_skips_on_inf_and_relinks_after_load(hip_lib, tmp_path):
    from nnuzoo_amd.training.fused_sgd import FusedSGD
    plans, cfg, dj = nnunet_plans(3, (32, 32, 32), batch_size=2)
    tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    batch = synthetic_batch(2, (32, 32, 32), tr._get_deep_supervision_scales(), seed=9)
    tr.train_step(batch)
    opt: FusedSGD = tr.optimizer
    before = {k: v.clone() for k, v in tr.network.state_dict().items()}
    mom_before = opt._flat_mom.clone()
    scale_before = float(tr.grad_scaler.get_scale())
    # poison one gradient element of the arena: the step must be skipped and the loss scale backed off
    arena = tr.network.grad_arena()
    arena[123] = float("inf")
    inv = tr.grad_scaler._scale.double().reciprocal().float()
    found = opt.fused_step(inv, 12)
    torch._amp_update_scale_(tr.grad_scaler._scale, tr.grad_scaler._growth_tracker, found, 2.0, 0.5, 2000)
    assert float(found) > 0
    for k, v in tr.network.state_dict().items():
        assert torch.equal(v, before[k]), k
    assert torch.equal(opt._flat_mom, mom_before)
    assert float(tr.grad_scaler.get_scale()) == 0.5 * scale_before
    # optimizer state survives a state_dict round trip and is re-linked into the flat buffer
    import copy
    sd = copy.deepcopy(opt.state_dict())          # what torch.load of a checkpoint hands over: fresh tensors
    mom_saved = opt._flat_mom.clone()
    opt.load_state_dict(sd)
    assert not opt._linked()
    opt._build(tr.network.grad_arena_layout(), mom_saved.device)
    assert opt._linked() and torch.equal(opt._flat_mom, mom_saved)   # restored buffers were copied into the flat tensor
    tr.train_step(batch)
    assert opt._linked()
    for p, off, _, _ in opt._links:
        assert opt.state[p]["momentum_buffer"].data_ptr() == opt._flat_mom.data_ptr() + 4 * off


def test_graph_replayed_steps_equal_eager_steps_bit_for_bit(hip_lib):
    """round 3: nnUNetTrainer.train_step replays forward + loss + backward as one hipGraph by default.  All kernels of the
    path are deterministic, so graph and eager training must agree to the last bit: losses of 4 steps, every parameter
    and the momentum buffers afterwards - with junk allocations between the steps (the captured graph owns its memory)"""
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer

    def run(graph):
        plans, cfg, dj = nnunet_plans(3, (64, 64, 64), batch_size=2)
        torch.manual_seed(0)
        tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
        tr.use_hip_graph = graph
        tr.initialize()
        losses = []
        for i in range(4):
            b = synthetic_batch(2, (64, 64, 64), tr._get_deep_supervision_scales(), seed=10 + i)      # a NEW batch each step
            b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
            losses.append(float(tr.train_step(b)["loss"]))
            junk = [torch.full((1 + 997 * k,), float("nan"), device="cuda") for k in range(50)]
            del junk
        mom = [tr.optimizer.state[p]["momentum_buffer"].clone() for p in tr.network.parameters() if p in tr.optimizer.state]
        return losses, [p.detach().clone() for p in tr.network.parameters()], mom, tr

    le, pe, me, _ = run(False)
    lg, pg, mg, tr = run(True)
    assert tr._graphed is not None and tr._graphed.graph is not None          # the graph path really ran
    assert le == lg, (le, lg)
    assert all(torch.equal(a, b) for a, b in zip(pe, pg))
    assert len(me) == len(mg) and all(torch.equal(a, b) for a, b in zip(me, mg))


def test_graph_and_eager_steps_alternate_with_fused_sgd(hip_lib):
    """FusedSGD reads the gradients from `network.grad_arena()`.  A graph replay runs no Python, so after an eager
    step in between the module still pointed at the EAGER pass's arena while the replay filled the captured one - the update
    was then computed from stale gradients.  GraphedForwardBackward re-registers the captured arena after every replay: steps
    alternating graph / eager / graph / ... must equal an all-eager run bit for bit (all kernels are deterministic)."""
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer
    from nnuzoo_amd.training.fused_sgd import FusedSGD

    def run(pattern):
        plans, cfg, dj = nnunet_plans(3, (32, 32, 32), batch_size=2)
        torch.manual_seed(0)
        tr = nnUNetTrainer(plans, cfg, 0, dj, device=torch.device("cuda"))
        tr.initialize()
        assert isinstance(tr.optimizer, FusedSGD)
        losses = []
        for i, g in enumerate(pattern):
            tr.use_hip_graph = g
            b = synthetic_batch(2, (32, 32, 32), tr._get_deep_supervision_scales(), seed=20 + i)
            b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
            losses.append(float(tr.train_step(b)["loss"]))
        return losses, [p.detach().clone() for p in tr.network.parameters()], tr

    le, pe, _ = run([False] * 6)
    lg, pg, tr = run([True, True, False, True, False, True])
    assert tr._graphed is not None and tr._graphed.graph is not None
    assert le == lg, (le, lg)
    assert all(torch.equal(a, b) for a, b in zip(pe, pg))
