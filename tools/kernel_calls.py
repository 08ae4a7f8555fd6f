"""Per-call durations of the kernels matching a pattern, in launch order, from a rocprofv3 --kernel-trace CSV directory:
   python tools/kernel_calls.py <dir> <substring> [last N calls]"""
import csv
import glob
import sys


def main():
    d, pat = sys.argv[1], sys.argv[2]
    last = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60],
                             r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
    rows.sort()
    for s, e, n, g, w in rows[-last:]:
        print(f"{(e - s) / 1e3:8.2f} us  grid {g} wg {w}  {n}")


if __name__ == "__main__":
    main()
