"""the dispatch sites of the zoo record which kernel family they took
(nnuzoo_amd/backends.py) instead of deciding behind a silent predicate, and the bench configurations run on the HIP
kernels everywhere except the documented ATen choices."""
import pytest
import torch

from nnuzoo_amd import backends as bk


def test_note_report_assert_cpu():
    bk.reset()
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.ReLU())
    bk.note(net[0], "hip-f32")
    bk.note(net[1], "library", why="test")
    bk.note(net[1], "hip", site="wgrad")
    assert net[0].backend == "hip-f32" and net[1].backend == "library" and net[1].backend_sites == {"wgrad": "hip"}
    rep = bk.report(net)
    assert rep == {"Linear": {"hip-f32": 1, "library": 1}, "Linear.wgrad": {"hip": 1}}
    with pytest.raises(AssertionError, match="library"):
        bk.assert_hip(net)
    net[1].backend = "hip-f16"
    assert bk.assert_hip(net) == {"Linear": {"hip-f32": 1, "hip-f16": 1}, "Linear.wgrad": {"hip": 1}}
    assert bk.COUNTS[("Linear", "forward", "library")] == 1
    assert "backend" not in net.state_dict() and len(net.state_dict()) == 4      # plain attributes: state_dict untouched


def _step(trainer_cls, size=128):
    from nnuzoo_amd.synthetic import nnunet_plans, synthetic_batch
    plans, cfg, dj = nnunet_plans(2, (size, size), batch_size=2)
    torch.manual_seed(0)
    tr = trainer_cls(plans, cfg, 0, dj, device=torch.device("cuda"))
    tr.initialize()
    b = synthetic_batch(2, (size, size), tr._get_deep_supervision_scales(), seed=3)
    b = {"data": b["data"].cuda(), "target": [t.cuda() for t in b["target"]]}
    tr.train_step(b)
    torch.cuda.synchronize()
    return tr


@pytest.mark.gpu
def test_m2net_bench_configuration_runs_on_hip(hip_lib):
    """bench.py's secondary leg (nnUNetTrainerM2Net, fp16 autocast): SS2D blocks fused, RSU4F on conv_box, every Linear on
    the token-major MFMA kernel"""
    from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerM2Net
    tr = _step(nnUNetTrainerM2Net, size=512)
    rep = bk.assert_hip(tr.network, allow=("TokenLinear",))
    assert rep["SS2D"] == {"hip": 80} and rep["RSU4F"] == {"hip": 3}
    # round 6: the eight MU stems (REBNCONV on its own) and the six 3x3 side heads on the tap-table conv kernels; the eight 1x1 patch
    # embeddings, the eight 1x1 stage outputs and the fuse convolution on the fp32 MFMA Linear / head1x1 kernels with fp16 activations
    assert rep["REBNCONV"] == {"hip": 8} and rep["Conv2d"] == {"hip-f32": 17, "hip": 6}, rep
    # 272 Linear layers: the ones on the fp16 token-major MFMA kernel carry the work (measured 175); the deep levels of the inner
    # U structures (fewer than 1024 tokens) and the feature counts that kernel does not take (512 / 1024-wide bottlenecks) run on
    # the fp32 MFMA kernels since round 5 (97, were on the GEMM library); only feature counts that are not multiples of 4 (none
    # in M2Net) would still reach the library
    tl = [m for m in tr.network.modules() if type(m).__name__ == "TokenLinear"]
    assert rep["TokenLinear"].get("hip-f16", 0) >= 170
    assert rep["TokenLinear"].get("library", 0) == 0 and rep["TokenLinear"].get("hip-f32", 0) >= 90, rep["TokenLinear"]


@pytest.mark.gpu
def test_swt2net_bench_configuration_runs_on_hip(hip_lib):
    """bench.py's SwT2Net leg (fp32 step): every Linear incl. the Mlp pairs on the fp32 MFMA kernels; round 6: the three RSU4F
    stages (depthwise-separable conv -> BatchNorm -> ReLU units), the residual stems and the 1x1 heads of the Swin U-net stages on
    csrc/sepconv32.hip + the fp32 MFMA Linear kernels (nnuzoo_amd/sepconv32.py).  What is left to ATen: the 1-channel stem of stage 1
    (a depthwise conv of ONE channel; its pointwise half has K = 1)"""
    from nnuzoo_amd.training.zoo_trainers import nnUNetTrainerSwT2Net
    tr = _step(nnUNetTrainerSwT2Net, size=512)
    rep = bk.assert_hip(tr.network)
    assert rep["TokenLinear"] == {"hip-f32": sum(type(m).__name__ == "TokenLinear" for m in tr.network.modules())}
    assert rep["RSU4F"] == {"hip-f32": 3}
    # stems of all eight stages on the HIP path (stage 1's 1-channel stem zero-padded to four channels); the 1x1 heads and patch embeddings of all eight stages on HIP;
    # no depthwise convolution left on ATen's kernels
    # (Conv2d: the 1x1 heads and the kernel = stride patch embeddings of the eight Swin U-net stages, both as token Linears)
    assert rep["Sequential"] == {"hip-f32": 8} and rep["Conv2d"] == {"hip-f32": 16}
    assert "_Conv2d" not in rep and "_Conv2d.wgrad" not in rep
    assert rep["Convolution"] == {"hip-f32": 7}          # side1 .. side6 + outconv on csrc/sepconv32.hip head1x1_*
