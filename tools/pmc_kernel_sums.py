"""Per-kernel sums of a rocprofv3 --pmc counter_collection.csv for the kernels whose name contains one of the needles.
    python tools/pmc_kernel_sums.py <counter_collection.csv> needle [needle ...] > profiles/<name>.json"""
import csv
import json
import sys
from collections import defaultdict


def main():
    path, needles = sys.argv[1], sys.argv[2:]
    sums = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if needles and not any(n in k for n in needles):
            continue
        short = k.split("(")[0][-70:]
        sums[short][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[short].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    out = {}
    for k, d in sums.items():
        e = {"dispatches": len(calls[k]), **{c: v for c, v in sorted(d.items())}}
        if d.get("SQ_BUSY_CYCLES") and d.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            e["mfma_busy_over_sq_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"]
        out[k] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
