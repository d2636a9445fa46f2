"""Dice within +-0.01 of the reference path on synthetic labels (BASELINE.json north_star; protocol of SURVEY.md §8d
at a reduced shape the CPU oracle finishes in seconds): same seeded weights, same synthetic batches, HIP step on the
GPU vs fp32 oracle step on the CPU, foreground Dice on 16 held-out patches."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_dice_within_one_percent(hip_lib):
    from dice_parity import run
    r = run(edge=32, steps=40, heldout=16)
    print(r)
    assert r["dice_oracle"] > 0.5, "the synthetic task must be learnt for the comparison to mean anything"
    assert r["abs_delta"] <= 0.01
    assert r["mask_agreement"] >= 0.99


def test_dice_at_64_cubed_against_the_committed_oracle_run(hip_lib):
    """The protocol at 64^3 / 100 steps with the oracle's side from tests/golden/dice_oracle_plainconv_64.json (the CPU oracle run
    of tools/dice_oracle_cpu.py in the build container; ~6 minutes of CPU there, seconds here) - what bench.py's primary `dice`
    runs live.  Same gates as above."""
    from dice_parity import run_vs_oracle
    fx = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dice_oracle_plainconv_64.json")
    r = run_vs_oracle(fx)
    print(r)
    assert r["dice_oracle"] > 0.5
    assert r["loss_abs_delta_step0"] <= 5e-3          # first step: same parameters, same batch (fp16 operands vs fp32)
    assert r["abs_delta"] <= 0.01
    assert r["mask_agreement"] >= 0.99
