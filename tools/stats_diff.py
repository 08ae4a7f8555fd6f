"""Per-step kernel time of two rocprofv3 --kernel-trace --stats runs, normalised by the calls of mask_match_kernel (one per step):
   python tools/stats_diff.py <dirA> <dirB>   -> kernels by |difference|, with calls per step."""
import csv
import glob
import sys


def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3) for r in csv.DictReader(open(f))}
    steps = next(c for n, (c, _) in rows.items() if "mask_match_kernel" in n)
    return {n: (c / steps, t / steps) for n, (c, t) in rows.items()}, steps


def main():
    a, sa = load(sys.argv[1])
    b, sb = load(sys.argv[2])
    print(f"steps {sa} / {sb};  kernel time per step {sum(t for _, t in a.values()):.0f} / {sum(t for _, t in b.values()):.0f} us;"
          f"  launches per step {sum(c for c, _ in a.values()):.0f} / {sum(c for c, _ in b.values()):.0f}")
    keys = sorted(set(a) | set(b), key=lambda k: -abs(b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1]))
    for k in keys[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
        ca, ta = a.get(k, (0, 0))
        cb, tb = b.get(k, (0, 0))
        print(f"{ta:8.1f} {tb:8.1f} {tb - ta:+8.1f}   x{ca:4.1f} x{cb:4.1f}  {k[:90]}")


if __name__ == "__main__":
    main()
