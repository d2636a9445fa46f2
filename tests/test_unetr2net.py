"""UNETR2Net (nnuzoo_amd/nets/unetr2net.py; reference nets/unetr2net.py).  The inner nets are monai's ViT + UNETR blocks
(absent here: PARITY UNPINNED, restated from monai 1.3); what is checked: structure / parameter naming as restated, the
attention core against the explicit softmax formula of monai's SABlock, whole-net shapes, gradients and a trainer step."""
import numpy as np
import pytest
import torch


def test_structure_cpu():
    from nnuzoo_amd.nets.unetr2net import UNETR2Net
    net = UNETR2Net(2, 1, 2, True, [512, 512])
    keys = list(net.state_dict())
    assert round(sum(p.numel() for p in net.parameters()) / 1e6, 1) == 137.6
    for k in ("stage1.rebnconvin.0.conv.weight", "stage1.vit.patch_embedding.position_embeddings",
              "stage1.vit.blocks.0.attn.qkv.weight", "stage1.vit.blocks.0.attn.out_proj.bias", "stage1.vit.norm.weight",
              "stage2.encoder2.blocks.0.1.conv1.conv.weight", "stage5d.decoder5.conv_block.conv3.conv.weight",
              "patch_merging5.reduction.weight", "concat_back_dim1d.bias", "side6.conv.bias", "outconv.conv.weight"):
        assert k in keys, k
    assert "stage1.vit.blocks.0.attn.qkv.bias" not in keys                       # qkv_bias=False
    assert net.stage1.vit.patch_embedding.position_embeddings.shape == (1, 1024, 96)   # 512^2 / 16^2 tokens
    assert net.stage1.vit.blocks[0].attn.head_dim == 8 and net.stage3.vit.blocks[0].attn.head_dim == 32


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,H,D", [(2, 1024, 12, 8), (1, 256, 12, 16), (2, 64, 12, 32), (3, 4, 12, 32), (1, 777, 3, 24),
                                     (2, 130, 5, 2), (1, 1024, 12, 32)])
def test_global_attention_vs_explicit_formula(hip_lib, B, L, H, D):
    """csrc/global_attention.hip (hand-written flash-style kernels, round 3) against the explicit float64 formula of monai's
    SABlock; ragged L (not a multiple of the 64-token blocks / 128-token workgroups), all head_dims of the zoo and odd ones;
    the backward has no atomics: two runs are bit-identical"""
    from nnuzoo_amd.global_attention import global_attention
    g = torch.Generator().manual_seed(L + D)
    qkv = torch.randn(B, L, 3, H, D, generator=g).cuda().requires_grad_(True)
    go = torch.randn(B, L, H * D, generator=g).cuda()
    scale = D ** -0.5
    o = global_attention(qkv, scale)
    o.backward(go)
    ref_in = qkv.detach().double().requires_grad_(True)
    q, k, v = (ref_in[:, :, i].transpose(1, 2) for i in range(3))               # (B, H, L, D)
    att = (torch.einsum("blxd,blyd->blxy", q, k) * scale).softmax(dim=-1)
    ref = torch.einsum("bhxy,bhyd->bhxd", att, v).transpose(1, 2).reshape(B, L, H * D)
    ref.backward(go.double())
    assert torch.allclose(o.double(), ref, rtol=2e-4, atol=2e-5)
    assert torch.allclose(qkv.grad.double(), ref_in.grad, rtol=2e-3, atol=2e-5)
    g1 = qkv.grad.clone()
    qkv.grad = None
    global_attention(qkv, scale).backward(go)
    assert torch.equal(qkv.grad, g1)
    with torch.autocast("cuda", dtype=torch.float16):                       # autocast steps hand over fp16 qkv
        oh = global_attention(qkv.detach().half(), scale)
    assert oh.dtype == torch.float16 and torch.allclose(oh.double(), ref.detach(), rtol=2e-2, atol=2e-2)


@pytest.mark.gpu
def test_whole_net_and_trainer_step(hip_lib):
    from nnuzoo_amd.nets.unetr2net import UNETR2Net
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerUNETR2Net
    prev = torch.backends.cudnn.enabled
    torch.backends.cudnn.enabled = False
    try:
        torch.manual_seed(0)
        net = UNETR2Net(2, 1, 2, True, [64, 64]).cuda()
        outs = net(torch.randn(2, 1, 64, 64, device="cuda"))
        assert [tuple(o.shape[2:]) for o in outs] == [(64, 64), (64, 64), (32, 32), (16, 16), (8, 8), (4, 4), (4, 4)]
        sum(o.float().pow(2).mean() for o in outs).backward()
        assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in net.parameters())
        del net, outs
        plans, cfg, dj = nnunet_plans(2, (64, 64), batch_size=2)
        tr = nnUNetTrainerUNETR2Net(plans, cfg, 0, dj, device=torch.device("cuda"))
        assert tr.num_epochs == 1000
        tr.initialize()
        b = synthetic_batch(2, (64, 64), tr._get_deep_supervision_scales(), seed=1)
        losses = [float(tr.train_step({"data": b["data"], "target": b["target"]})["loss"]) for _ in range(3)]
        assert all(np.isfinite(losses))
    finally:
        torch.backends.cudnn.enabled = prev


def test_attention_outside_the_kernel_set_runs_on_the_library_and_says_so():
    """monai's UNETR / ViT defaults (hidden 768, 12 heads = head_dim 64) and CPU tensors must still run - on
    torch's SDPA, with the choice recorded on the module (never a silent fallback, never a raise).  CPU test."""
    from nnuzoo_amd import backends
    from nnuzoo_amd.nets.unetr2net import SABlock
    torch.manual_seed(0)
    blk = SABlock(768, 12)
    x = torch.randn(2, 10, 768)
    y = blk(x)
    qkv = blk.qkv(x).view(2, 10, 3, 12, 64)
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
    ref = blk.out_proj(((q @ k.transpose(-1, -2) * blk.scale).softmax(-1) @ v).transpose(1, 2).reshape(2, 10, 768))
    assert torch.allclose(y, ref, atol=1e-5)
    assert backends.report(blk) == {"SABlock.attention": {"library": 1}}
