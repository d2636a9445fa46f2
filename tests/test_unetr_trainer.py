"""nnUNetTrainerUNETR (reference training/nnUNetTrainer/nnUNetTrainerUNETR.py): monai's UNETR restated on this package's ViT /
UNETR blocks (`nets/unetr2net.py MonaiUNETR`; monai absent: structure and step tested, parity unpinned), patch size rounded up to
the ViT's 16-voxel patches, fp32 step, one output."""
import numpy as np
import pytest
import torch


def test_unetr_structure_patch_rounding_and_namespace():
    from nnuzoo_amd.nets.unetr2net import MonaiUNETR
    from nnuzoo_amd.synthetic import nnunet_plans
    from nnuzoo_amd.training.nnUNetTrainer import nnUNetTrainer
    from nnunetv2.training.nnUNetTrainer.nnUNetTrainerUNETR import nnUNetTrainerUNETR
    assert issubclass(nnUNetTrainerUNETR, nnUNetTrainer)
    net = MonaiUNETR(1, 3, [32, 48], spatial_dims=2)
    keys = list(net.state_dict().keys())
    # monai's child names and depths: 12 transformer blocks, skips after blocks 3 / 6 / 9, encoder2 with two up-sampling stages
    assert keys[0] == "vit.patch_embedding.position_embeddings" and keys[-1] == "out.conv.conv.bias"
    assert sum(k.startswith("vit.blocks.") and k.endswith("attn.qkv.weight") for k in keys) == 12
    assert net.out_indices == [3, 6, 9] and not net.add_last and not any(k.startswith("rebnconvin") for k in keys)
    assert any(k.startswith("encoder2.blocks.1.") for k in keys) and not any(k.startswith("encoder4.blocks.0.") for k in keys)
    assert net.state_dict()["vit.patch_embedding.position_embeddings"].shape == (1, 2 * 3, 768)
    plans, cfg, dj = nnunet_plans(2, (40, 70), batch_size=2)
    tr = nnUNetTrainerUNETR(plans, cfg, 0, dj, device=torch.device("cpu"))
    assert tr.configuration_manager.patch_size == [48, 80]          # 40 -> 48, 70 -> 80: multiples of the 16-voxel patch
    assert plans["configurations"][cfg]["patch_size"] == [48, 80]
    assert tr.grad_scaler is None and tr._get_deep_supervision_scales() is None and tr.weight_decay == 0.01
    plans, cfg, dj = nnunet_plans(3, (32, 16, 64), batch_size=1)
    assert nnUNetTrainerUNETR(plans, cfg, 0, dj, device=torch.device("cpu")).configuration_manager.patch_size == [32, 16, 64]


@pytest.mark.gpu
def test_unetr_trainer_steps_2d(hip_lib):
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerUNETR
    plans, cfg, dj = nnunet_plans(2, (64, 64), batch_size=2)
    tr = nnUNetTrainerUNETR(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    b = synthetic_batch(2, (64, 64), [[1.0, 1.0]], seed=3)
    b = {"data": b["data"], "target": b["target"][0]}
    losses = [float(tr.train_step(b)["loss"]) for _ in range(6)]
    assert all(np.isfinite(l) for l in losses) and losses[-1] < losses[0]
    out = tr.validation_step(b)
    assert np.isfinite(float(out["loss"])) and out["tp_hard"].shape[0] >= 1
