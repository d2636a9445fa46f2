"""The published measurement artefacts are self-consistent (CPU, no GPU needed): the HBM traffic figure bench.py reports is
what tools/pmc_traffic.py derives from the committed PMC passes, and the bench line of record agrees with the committed
rocprofv3 kernel statistics on the dominant kernel's average launch time."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def test_hbm_traffic_json_is_derived_from_the_committed_pmc_passes():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_traffic import per_kernel
    f, nf = per_kernel(os.path.join(PROF, "r06_pmc_fetch_size.csv"), "FETCH_SIZE", "conv_box_kernel")
    w, nw = per_kernel(os.path.join(PROF, "r06_pmc_write_size.csv"), "WRITE_SIZE", "conv_box_kernel")
    assert nf == nw and nf > 0
    derived = (2 * f + w) * 1024 / nf          # gfx950: FETCH_SIZE counts 64 B per 128-B request, units of KiB
    pub = json.load(open(os.path.join(PROF, "conv_box_kernel_hbm_traffic.json")))
    assert abs(pub["hbm_bytes_per_launch"] - derived) <= 1e-6 * derived
    assert pub["launches_sampled"] == nf
    # traffic above the algorithmic bytes, but within 2x (input + output of the tile once ~ 160 MB per launch; since round 3
    # the data-gradient launches also read the layer-below tile for the fused norm-backward reductions: + ~25 MB; round 4:
    # raw inputs + tables instead of activated tensors - same volume)
    assert 1.6e8 < derived < 3.2e8


def test_bench_line_agrees_with_the_kernel_stats_file():
    line = json.loads(open(os.path.join(PROF, "r06_bench_n1.json")).read().strip().splitlines()[-1])
    roof = line["roofline"]
    assert line["unit"] == "patches/s" and line["n_gpus"] == 1 and roof["bound"] == "mfma"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    rows = [r for r in csv.DictReader(open(os.path.join(PROF, "r06_bench_n1_kernel_stats.csv")))
            if "conv_box_kernel" in r["Name"]]
    calls = sum(int(r["Calls"]) for r in rows)
    avg_us = sum(float(r["TotalDurationNs"]) for r in rows) / calls / 1e3
    # live HIP-event average of bench.py vs rocprofv3's average for the same kernel family (different runs / boxes)
    assert abs(avg_us - roof["avg_launch_us"]) <= 0.10 * roof["avg_launch_us"], (avg_us, roof["avg_launch_us"])
    assert calls % roof["launches_per_step"] == 0
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0
    # round 6: the zoo legs' headline numbers are top-level scalars of the contract line too
    assert line["secondary_value"] == line["secondary"]["value"] and line["swt2net_value"] == line["swt2net"]["value"]
    assert line["roofline_frac"] == roof["frac"] and "HBM-resident" in line["config"]["workload"]
