"""Per replayed step of a kernel trace: span, kernel time on the step's chain, idle time on the chain, and where the idle time sits
(gap before which kernel).  usage: step_gaps.py <trace dir> [n last steps]"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = list(csv.DictReader(open(f)))
def nm(r):
    return r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r), r["Queue_Id"]) for r in rows)
starts = [e[0] for e in ev if "rc_stats_kernel" in e[2]]
mainq = [e[3] for e in ev if "rc_stats_kernel" in e[2]][-1]
for s0, s1 in list(zip(starts[:-1], starts[1:]))[-n:]:
    ks = [e for e in ev if s0 <= e[0] < s1 and e[3] == mainq]
    busy, idle, gaps = s0, 0.0, []
    for s, e, name, q in ks:
        if s > busy:
            g = (s - busy) / 1e3
            idle += g
            if g > 4:
                gaps.append((round(g), name[:28]))
        busy = max(busy, e)
    idle += max(0, s1 - busy) / 1e3
    if s1 - busy > 4000:
        gaps.append((round((s1 - busy) / 1e3), "<next step>"))
    ktime = sum(e - s for s, e, _, _ in ks) / 1e3
    small = [(e - s) / 1e3 for s, e, _, _ in ks if e - s < 8000]
    print(f"span {(s1 - s0) / 1e3:7.1f}  chain kernels {len(ks):3d} time {ktime:7.1f}  idle {idle:6.1f}  <8us: {len(small)} = {sum(small):5.1f}  gaps {gaps}")
