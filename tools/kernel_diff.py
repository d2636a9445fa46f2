"""diff two tools/kernel_summary.py outputs by kernel name: ms per step (given steps in each window) - usage: kernel_diff.py A.txt B.txt"""
import re
import sys


def load(path):
    tot, span = {}, None
    for line in open(path):
        m = re.match(r"kernels (\d+)\s+busy ([\d.]+) ms\s+span ([\d.]+) ms", line)
        if m:
            span = float(m.group(3))
        m = re.match(r"\s+([\d.]+) ms\s+(\d+) calls\s+([\d.]+) us\s+(.*)", line)
        if m:
            name = re.sub(r"\(.*", "", m.group(4)).strip()[:90]
            t = tot.setdefault(name, [0.0, 0])
            t[0] += float(m.group(1))
            t[1] += int(m.group(2))
    return tot, span


a, sa = load(sys.argv[1])
b, sb = load(sys.argv[2])
na = a.get("nnz::adamw_kernel", [0, 1])[1] or 1
nb = b.get("nnz::adamw_kernel", [0, 1])[1] or 1
# the window holds a fractional number of steps: normalise by the window's span against the bench's ms per step if given
fa = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
fb = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
rows = []
for k in set(a) | set(b):
    ma, ca = a.get(k, [0.0, 0])
    mb, cb = b.get(k, [0.0, 0])
    rows.append((mb / fb - ma / fa, k, ma / fa, ca / fa, mb / fb, cb / fb))
rows.sort(key=lambda r: -abs(r[0]))
print(f"{'delta ms/step':>13s}  {'A ms':>8s} {'A calls':>8s}  {'B ms':>8s} {'B calls':>8s}  kernel")
for d, k, ma, ca, mb, cb in rows[:45]:
    print(f"{d:13.3f}  {ma:8.3f} {ca:8.1f}  {mb:8.3f} {cb:8.1f}  {k}")
print("sum of deltas", round(sum(r[0] for r in rows), 3))
