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
