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
