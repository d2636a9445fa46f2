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
    # every kernel that is NOT this package's (no `nnz` namespace in the name), by family - the whole list, not the top N:
    # what a library (MIOpen / rocBLAS / hipBLASLt / CK) or ATen still contributes to the profiled steps
    def family(name):
        for key, fam in (("Cijk_", "rocBLAS / hipBLASLt (Cijk_*)"), ("igemm", "MIOpen igemm"), ("miopen", "MIOpen"),
                         ("MIOpen", "MIOpen"), ("ck::", "composable_kernel"), ("Gridwise", "composable_kernel"),
                         ("rccl", "RCCL"), ("nccl", "RCCL"), ("at::native", "ATen"), ("__amd_rocclr", "runtime copy / fill")):
            if key in name:
                return fam
        return "other"
    foreign = defaultdict(lambda: [0, 0, []])
    for name, (n, t) in agg.items():
        if "nnz" in name:
            continue
        f = foreign[family(name)]
        f[0] += n
        f[1] += t
        f[2].append((t, n, name))
    ftot = sum(v[1] for v in foreign.values())
    print(f"kernels of other origin than this package: {ftot / 1e6:.3f} ms of {tot / 1e6:.2f} ms busy ({100.0 * ftot / max(tot, 1):.1f} %)")
    for fam, (n, t, members) in sorted(foreign.items(), key=lambda kv: -kv[1][1]):
        print(f"{t / 1e6:9.3f} ms {n:6d} calls  {fam}  ({len(members)} distinct)")
        for mt, mn, name in sorted(members, reverse=True)[:6 if fam == "ATen" else 40]:
            print(f"      {mt / 1e6:9.3f} ms {mn:6d} calls  {name[:100]}")
    # idle gaps between consecutive kernels (host-bound stretches): histogram + the kernels that follow the largest gaps
    gaps = []
    end = int(rows[0]["End_Timestamp"])
    for prev, r in zip(rows, rows[1:]):
        st = int(r["Start_Timestamp"])
        if st > end:
            gaps.append((st - end, prev["Kernel_Name"], r["Kernel_Name"]))
        end = max(end, int(r["End_Timestamp"]))
    tot_gap = sum(g[0] for g in gaps)
    print(f"idle gaps: {len(gaps)} totalling {tot_gap / 1e6:.2f} ms; "
          f">100us: {sum(g[0] for g in gaps if g[0] > 1e5) / 1e6:.2f} ms in {sum(1 for g in gaps if g[0] > 1e5)}; "
          f"10-100us: {sum(g[0] for g in gaps if 1e4 < g[0] <= 1e5) / 1e6:.2f} ms in {sum(1 for g in gaps if 1e4 < g[0] <= 1e5)}; "
          f"<10us: {sum(g[0] for g in gaps if g[0] <= 1e4) / 1e6:.2f} ms")
    after = defaultdict(lambda: [0, 0])
    for g, pn, nn in gaps:
        after[nn[:70]][0] += 1
        after[nn[:70]][1] += g
    print("gap time by FOLLOWING kernel:")
    for name, (n, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"{t / 1e6:9.3f} ms {n:6d} gaps {t / n / 1e3:8.1f} us  -> {name}")
    for g, pn, nn in sorted(gaps, reverse=True)[:8]:
        print(f"  gap {g / 1e3:9.1f} us  after {pn[:50]}  before {nn[:50]}")


if __name__ == "__main__":
    main()
