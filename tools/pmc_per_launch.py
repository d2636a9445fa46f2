"""Per-launch HBM-side traffic of one kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) and, optionally, a
kernel trace of the same command: the last `n` launches (one training step), each with its grid, fetched / written MB and duration.
    python tools/pmc_per_launch.py fetch.csv write.csv conv_box_kernel 49 [kernel_trace.csv]
gfx950: FETCH_SIZE counts 64 B per 128-B request in units of KiB -> bytes = 2 * value * 1024; WRITE_SIZE: value * 1024
(MI355X_MICROARCH.md, HBM / rocprofv3 section; tools/pmc_traffic.py uses the same corrections)."""
import csv
import sys


def rows(path, name, counter):
    return [r for r in csv.DictReader(open(path)) if name in r["Kernel_Name"] and r["Counter_Name"] == counter]


def main():
    f = rows(sys.argv[1], sys.argv[3], "FETCH_SIZE")
    w = rows(sys.argv[2], sys.argv[3], "WRITE_SIZE")
    n = int(sys.argv[4])
    f, w = f[-n:], w[-n:]
    dur = None
    if len(sys.argv) > 5:
        tr = [r for r in csv.DictReader(open(sys.argv[5])) if sys.argv[3] in r["Kernel_Name"]][-n:]
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
    tf = tw = 0.0
    for i, (a, b) in enumerate(zip(f, w)):
        assert a["Grid_Size"] == b["Grid_Size"], (i, a["Grid_Size"], b["Grid_Size"])
        fm, wm = 2 * float(a["Counter_Value"]) * 1024 / 1e6, float(b["Counter_Value"]) * 1024 / 1e6
        tf, tw = tf + fm, tw + wm
        name = a["Kernel_Name"].split("conv_box_kernel")[-1][:58] if "conv_box_kernel" in a["Kernel_Name"] else a["Kernel_Name"][:58]
        extra = ""
        if dur:
            extra = f"  {dur[i]:8.1f} us  {(fm + wm) / dur[i]:5.2f} TB/s"
        print(f"{i:3d} wgs {int(a['Grid_Size']) // 256:6d}  fetch {fm:8.1f} MB  write {wm:8.1f} MB{extra}  {name}")
    print(f"total fetch {tf:.0f} MB, write {tw:.0f} MB over {len(f)} launches -> {(tf + tw) / len(f):.1f} MB per launch")


if __name__ == "__main__":
    main()
