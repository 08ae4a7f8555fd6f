"""Per-kernel time per step of two rocprofv3 kernel traces (last 10 steps each, split at fps_kernel<256 launches)."""
import csv, glob, sys, collections
def load(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    st = [e[0] for e in ev if "fps_kernel<256" in e[2]]
    s0, s1, n = st[-11], st[-1], 10
    acc = collections.Counter(); cnt = collections.Counter()
    for a, b, k in ev:
        if s0 <= a < s1:
            acc[k[:90]] += (b - a) / 1e3 / n; cnt[k[:90]] += 1 / n
    return acc, cnt, (s1 - s0) / 1e3 / n
a, ca, ta = load(sys.argv[1]); b, cb, tb = load(sys.argv[2])
print("step us", round(ta), round(tb), " kernel sums", round(sum(a.values())), round(sum(b.values())))
for k in sorted(set(a) | set(b), key=lambda k: -abs(b.get(k, 0) - a.get(k, 0)))[:25]:
    print(f"{a.get(k, 0):8.1f} {b.get(k, 0):8.1f} {b.get(k, 0) - a.get(k, 0):+7.1f}  x{cb.get(k, ca.get(k, 0)):.0f} {k[:70]}")
