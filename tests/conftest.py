import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI library; GPU tests fail loudly (not skip) when it is missing."""
    from nnuzoo_amd import _lib
    return _lib.load()


@pytest.fixture
def force_scan_gen2():
    """Context-manager factory: inside `with force_scan_gen2(clb):` every selective-scan / cross-scan call that generation
    2 (channels on the lanes, csrc/ss2d_scan_rl.hpp) CAN take does take it, whatever its size (by default only calls of
    >= 2 M row-steps do - the bench shapes, which no golden reaches); yields a callable returning the number of calls that
    took the generation-2 kernels since entry."""
    import contextlib

    @contextlib.contextmanager
    def _force(clb: int = 0):
        from nnuzoo_amd._lib import call, load
        lib = load()
        saved = [lib.nnz_scan_tuning_get(k) for k in range(3)]
        call("nnz_scan_tuning", 0, 1)
        call("nnz_scan_tuning", 1, clb)
        call("nnz_scan_tuning", 2, 0)
        before = lib.nnz_scan_tuning_get(3)
        try:
            yield lambda: lib.nnz_scan_tuning_get(3) - before
        finally:
            for k, v in enumerate(saved):
                call("nnz_scan_tuning", k, v)
    return _force


# ---- collection order ---------------------------------------------------------------------------------------
# The driver runs `pytest -x`: a failure hides everything collected behind it.  Parity evidence (golden / oracle comparisons) is
# therefore collected FIRST, self-consistency properties (determinism, fused-vs-composed, dispatch reports) second, and the
# statistical protocols (training trajectories, Dice protocols, loss-descent checks - the only tests whose outcome depends on a
# chaotic fp16 trajectory) LAST.  Within a tier the default (alphabetical / definition) order is kept.
_TIER_BY_FILE = {
    # tier 1: properties of the HIP path against itself / host plumbing
    "test_ss2d_cross_scan_gpu.py": 1, "test_determinism_gpu.py": 1, "test_zoo_determinism_gpu.py": 1, "test_backends.py": 1,
    "test_graph_replay_gpu.py": 1, "test_param_shadow_gpu.py": 1, "test_ddp_rccl_gpu.py": 1, "test_two_stage_wgrads_gpu.py": 1,
    "test_bench_launch.py": 1, "test_profiles_consistency.py": 1, "test_device_augment_gpu.py": 1,
    # tier 2: statistical protocols
    "test_zoo_trajectory_gpu.py": 2, "test_dice_parity_gpu.py": 2, "test_dice_parity_zoo_gpu.py": 2,
}
_TIER_BY_NAME = {
    "test_ssnd2net_training_descends_once_the_loss_scale_has_settled": 2,
    "test_ssnd2net_fp32_step_applies_its_updates_at_a_gradient_norm_of_1e9": 2,
    "test_m2net_steps_call_no_torch_convolution_or_batchnorm_module": 1, "test_side_head_3x3_repeats_bit_for_bit": 1,
    "test_adjoint_identity_and_repeatability": 1,
}


def collection_tier(item) -> int:
    name = item.name.split("[")[0]
    if name in _TIER_BY_NAME:
        return _TIER_BY_NAME[name]
    return _TIER_BY_FILE.get(os.path.basename(str(item.fspath)), 0)


def pytest_collection_modifyitems(config, items):
    items.sort(key=collection_tier)          # stable: keeps the order inside a tier
