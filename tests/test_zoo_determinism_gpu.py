"""Run-to-run bit-identity of a whole M2NetP training step.  Round 5 removed the float atomics of the SS2D^2Net
step: the scan backward writes per-workgroup slabs and folds them in a fixed order (csrc/selective_scan.hip, ss2d_scan_rl.hpp), the
weight gradients of the fp16 token Linears and of x_proj leave their kernels as per-workgroup partial blocks folded by the pass's
grouped launches (csrc/token_linear.hip, ss2d_xproj.hip), the depthwise conv + SiLU backward takes its two-stage form
(the default since the end of round 6: it costs nothing any more; NNZ_TWO_STAGE_WGRADS=0 restores the atomics), LayerNorm and the optimizer tail were fixed-point already.
Two trainers built from the same seed and fed the same batches must therefore hold the SAME parameters, bit for bit, after
several steps - eager and replayed as a hipGraph.  What is NOT ours stays outside the claim: the library convolutions (the 1 x 1 side
heads and the fuse convolution; MIOpen picks its solver per process, profiles/r04_zoo_module_determinism.txt) are replaced by ATen's
kernels for the test, which is what `NNZ_LIBRARY_DETERMINISTIC=2` selects in the trainers."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(graph: bool, steps: int = 4, trainer: str = "nnUNetTrainerM2NetP"):
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training import zoo_trainers
    plans, cfg, dj = nnunet_plans(2, (128, 128), batch_size=2)
    torch.manual_seed(0)
    tr = getattr(zoo_trainers, trainer)(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    tr.use_hip_graph = graph
    scales = tr._get_deep_supervision_scales()
    losses = []
    torch.manual_seed(1)            # DropPath draws
    for it in range(steps):
        b = synthetic_batch(2, (128, 128), scales, seed=100 + it)
        b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
        losses.append(float(tr.train_step(b)["loss"]))
    torch.cuda.synchronize()
    net = tr.network.module if hasattr(tr.network, "module") else tr.network
    return losses, {n: p.detach().clone() for n, p in net.named_parameters()}


@pytest.mark.parametrize("graph", [False, True])
def test_m2netp_training_steps_are_bit_reproducible(hip_lib, graph, monkeypatch):
    from nnuzoo_amd import token_linear
    assert token_linear.TWO_STAGE       # the default since the end of round 6: depthwise conv + SiLU weight gradient as partial rows + fold
    with torch.backends.cudnn.flags(enabled=False):
        la, pa = _run(graph)
        lb, pb = _run(graph)
    assert la == lb, (la, lb)
    diff = [n for n in pa if not torch.equal(pa[n], pb[n])]
    assert not diff, (len(diff), diff[:8])


def _run_swt(graph: bool, steps: int = 4):
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerSwT2Net
    plans, cfg, dj = nnunet_plans(2, (128, 128), batch_size=2)
    torch.manual_seed(0)
    tr = nnUNetTrainerSwT2Net(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    tr.use_hip_graph = graph
    scales = tr._get_deep_supervision_scales()
    losses = []
    torch.manual_seed(1)            # DropPath draws
    for it in range(steps):
        b = synthetic_batch(2, (128, 128), scales, seed=100 + it)
        b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
        losses.append(float(tr.train_step(b)["loss"]))
    torch.cuda.synchronize()
    return losses, {n: p.detach().clone() for n, p in tr.network.named_parameters()}


@pytest.mark.parametrize("graph", [False, True])
def test_swt2net_training_steps_are_bit_reproducible_in_the_default_mode(hip_lib, graph):
    """round 6: with the RSU4F stages, the stage stems / heads / patch embeddings and the side heads on csrc/sepconv32.hip no
    convolution of the SwT2Net step goes to MIOpen any more, so the step needs no `NNZ_LIBRARY_DETERMINISTIC` switch: two trainers
    built from one seed hold bit-identical parameters after four steps in the DEFAULT library mode (what is left on a library - the
    two small matmuls of the bilinear up-sampling's backward - is a plain GEMM)"""
    from nnuzoo_amd import backends as bk
    la, pa = _run_swt(graph)
    lb, pb = _run_swt(graph)
    assert la == lb, (la, lb)
    diff = [n for n in pa if not torch.equal(pa[n], pb[n])]
    assert not diff, (len(diff), diff[:8])


@pytest.mark.parametrize("graph", [False, True])
def test_m2net_training_steps_are_bit_reproducible_in_the_default_mode(hip_lib, graph, monkeypatch):
    """round 6, the net of bench.py's secondary leg: the MU stems, 1x1 patch embeddings / stage outputs, 3x3 side heads and the fuse
    convolution run on this package's kernels (nnuzoo_amd/rebnconv.py unit_nchw / head3x3, sepconv32.py pointwise_tokens / head1x1),
    the REBNCONV batch statistics are fixed-point sums and the depthwise conv + SiLU weight gradient takes its two-stage form by
    default - no convolution of the M2Net step reaches MIOpen, no float atomic is left in it, and NOTHING is switched for this test:
    same seed, same batches, bit-identical parameters after four steps in the trainer's default mode, eager and replayed.
    (M2NetP keeps its 16-channel RSU4F units on the library - the tap-table kernels work on 32-channel blocks - hence the ATen
    switch in its test above.)"""
    from nnuzoo_amd import token_linear
    assert token_linear.TWO_STAGE       # nothing is switched for this test: the trainer's default mode
    la, pa = _run(graph, trainer="nnUNetTrainerM2Net")
    lb, pb = _run(graph, trainer="nnUNetTrainerM2Net")
    assert la == lb, (la, lb)
    diff = [n for n in pa if not torch.equal(pa[n], pb[n])]
    assert not diff, (len(diff), diff[:8])
