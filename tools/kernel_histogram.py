"""Per-kernel duration distribution of a rocprofv3 --kernel-trace csv (last `frac` of the trace): calls, min / median / p90 /
max us and total ms, with the launch's workgroup count at the median - tells latency-bound families (flat distribution at any
size) from bandwidth-bound ones.   python tools/kernel_histogram.py <kernel_trace.csv> [frac=0.5] [top=25]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[int(len(rows) * (1 - frac)):]
    by = defaultdict(list)
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        wgs = 1
        for ax in ("X", "Y", "Z"):
            g, w = int(r.get(f"Grid_Size_{ax}", r.get(f"Grid_Size", 1)) or 1), int(r.get(f"Workgroup_Size_{ax}", 1) or 1)
            wgs *= max(1, g // max(1, w))
        by[r["Kernel_Name"].split("(")[0][-60:]].append((d, wgs))
    tot = sorted(by.items(), key=lambda kv: -sum(d for d, _ in kv[1]))[:top]
    print(f"{'kernel':60s} {'calls':>6s} {'total ms':>9s} {'min':>7s} {'med':>7s} {'p90':>7s} {'max':>8s}  wgs(min/med/max)")
    for k, v in tot:
        v.sort()
        ds = [d for d, _ in v]
        ws = sorted(w for _, w in v)
        print(f"{k:60s} {len(v):6d} {sum(ds) / 1e3:9.2f} {ds[0]:7.1f} {ds[len(ds) // 2]:7.1f} {ds[int(len(ds) * 0.9)]:7.1f} "
              f"{ds[-1]:8.1f}  {ws[0]}/{ws[len(ws) // 2]}/{ws[-1]}")


if __name__ == "__main__":
    main()
