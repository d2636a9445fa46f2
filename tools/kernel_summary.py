"""Aggregate a rocprofv3 --kernel-trace csv by kernel name: calls, total ms, average us (top N).
Usage: python tools/kernel_summary.py <kernel_trace.csv> [topN] [skip_fraction]
skip_fraction: ignore the first part of the trace (warm-up), e.g. 0.5"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[int(len(rows) * skip):]
    agg = defaultdict(lambda: [0, 0])
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += d
    tot = sum(v[1] for v in agg.values())
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    print(f"kernels {len(rows)}  busy {tot / 1e6:.2f} ms  span {span / 1e6:.2f} ms")
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{t / 1e6:9.3f} ms {n:6d} calls {t / n / 1e3:9.2f} us  {name[:110]}")


if __name__ == "__main__":
    main()
