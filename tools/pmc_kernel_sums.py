"""Per-kernel sums of a rocprofv3 --pmc counter_collection.csv for the kernels whose name contains one of the needles, plus one "TOTAL <needle>"
entry per needle (all instantiations of a kernel template together).
    python tools/pmc_kernel_sums.py <counter_collection.csv> needle [needle ...] > profiles/<name>.json

Derived figures (units per /opt/skills/guides/MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* count QUAD-cycles, SQ_VALU_MFMA_BUSY_CYCLES
counts cycles, = 32 per 32x32x16 f16 MFMA):
  mfma_busy_cycles_per_mfma   SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA
  mfma_busy_over_wave_cycles  SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES): the share of a wave's lifetime in which one of ITS matrix
                              instructions occupies the pipe (x resident waves per SIMD = the SIMD's matrix-pipe utilisation while the
                              kernel's waves are resident)
  wait_over_wave_cycles       SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (both quad-cycles)
  valu_per_mfma               SQ_INSTS_VALU / SQ_INSTS_MFMA
"""
import csv
import json
import sys
from collections import defaultdict


def short_name(k: str) -> str:
    k = k.split("(")[0]
    for pre in ("void ", "nnz::"):
        if k.startswith(pre):
            k = k[len(pre):]
    return k[:110]


def derived(d):
    e = {}
    mf, wv = d.get("SQ_INSTS_MFMA", 0.0), d.get("SQ_WAVE_CYCLES", 0.0)
    if mf:
        e["mfma_busy_cycles_per_mfma"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / mf
        e["valu_per_mfma"] = d.get("SQ_INSTS_VALU", 0.0) / mf
    if wv:
        e["mfma_busy_over_wave_cycles"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * wv)
        if "SQ_WAIT_INST_ANY" in d:
            e["wait_over_wave_cycles"] = d["SQ_WAIT_INST_ANY"] / wv
    return e


def main():
    path, needles = sys.argv[1], sys.argv[2:]
    sums = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        hit = [n for n in needles if n in k]
        if needles and not hit:
            continue
        disp = r.get("Dispatch_Id", r.get("Correlation_Id", ""))
        for key in [short_name(k)] + [f"TOTAL {n}" for n in hit]:
            sums[key][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[key].add(disp)
    out = {}
    for k, d in sorted(sums.items(), key=lambda kv: (not kv[0].startswith("TOTAL"), -kv[1].get("SQ_WAVE_CYCLES", 0.0))):
        out[k] = {"dispatches": len(calls[k]), **{c: v for c, v in sorted(d.items())}, **derived(d)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
