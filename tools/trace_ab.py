"""Per-kernel launches and time per replayed step of two kernel traces (A | B), kernels that differ by more than a threshold.
usage: trace_ab.py <trace dir A> <trace dir B> [min |delta| us]"""
import collections, csv, glob, sys
def load(d):
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    nm = lambda r: r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:58]
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r)) for r in rows)
    st = [e[0] for e in ev if "rc_stats" in e[2]]
    spans = [b - a for a, b in zip(st[:-1], st[1:])][-8:]
    med = sorted(spans)[len(spans) // 2]
    out, n = collections.OrderedDict(), 0
    for s0, s1 in list(zip(st[:-1], st[1:]))[-8:]:
        if s1 - s0 > 1.15 * med:
            continue            # an eagerly launched (profiled) step
        n += 1
        for e in ev:
            if s0 <= e[0] < s1:
                out.setdefault(e[2], []).append((e[1] - e[0]) / 1e3)
    return {k: (len(v) / n, sum(v) / n) for k, v in out.items()}, n, med / 1e3
a, na, ma = load(sys.argv[1]); b, nb, mb = load(sys.argv[2])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5
print(f"steps {na} | {nb}; median span {ma:.1f} | {mb:.1f} us")
tot = 0
for n in list(a.keys()) + [k for k in b if k not in a]:
    ca, sa = a.get(n, (0, 0)); cb, sb = b.get(n, (0, 0))
    if abs(sa - sb) > thr:
        print(f"{n:60s} {ca:4.1f} {sa:7.1f} | {cb:4.1f} {sb:7.1f}  {sb - sa:+7.1f}")
    tot += sb - sa
print(f"total kernel time per step: {tot:+.1f} us")
