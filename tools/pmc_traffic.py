"""HBM traffic per conv_box launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected separately, as
MI355X_MICROARCH.md §HBM prescribes).  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB and the
gfx950 FETCH_SIZE counts 64 B per 128-B request.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --steps 2 --warmup 1 \
              --no-cpu-baseline --no-launch-timer
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 bench.py ... (same)
    python tools/pmc_traffic.py <fetch counter csv> <write counter csv> [kernel substring[,substring...] [count substring]] > profiles/<name>.json

Several comma-separated substrings: the bytes of all matching kernels are summed (a cross-scan backward CALL is a summary, a carry,
a final and a finalize kernel); `count substring` then names the kernel whose launches count the calls (default: all matches).
"""
import csv
import json
import sys


def per_kernel(path, counter, needle, count_needle=None):
    needles = needle.split(",")
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in needles):
            tot += float(r["Counter_Value"])
            if count_needle is None or any(k in r["Kernel_Name"] for k in count_needle.split(",")):
                n += 1
    return tot, n


def main():
    fetch_csv, write_csv = sys.argv[1], sys.argv[2]
    needle = sys.argv[3] if len(sys.argv) > 3 else "conv_box_kernel"
    count_needle = (sys.argv[4] or None) if len(sys.argv) > 4 else None
    cmd = sys.argv[5] if len(sys.argv) > 5 else "bench.py --steps 2 --warmup 1"
    f, nf = per_kernel(fetch_csv, "FETCH_SIZE", needle, count_needle)
    w, nw = per_kernel(write_csv, "WRITE_SIZE", needle, count_needle)
    assert nf == nw and nf > 0, (nf, nw)
    print(json.dumps({
        "kernel": f"{needle} (all launches of the profiled steps" + (f"; per launch of {count_needle})" if count_needle else ")"),
        "hbm_bytes_per_launch": (2 * f + w) * 1024 / nf,
        "fetch_bytes_per_launch": 2 * f * 1024 / nf,
        "write_bytes_per_launch": w * 1024 / nf,
        "method": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `{cmd}`; "
                  "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts 64 B per 128-B "
                  "request, MI355X_MICROARCH.md §HBM), averaged over the launches",
        "launches_sampled": nf}, indent=1))


if __name__ == "__main__":
    main()
