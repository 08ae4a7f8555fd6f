"""Per kernel name: mean duration when it ran alone on the device vs. while a side-stream kernel (FPS / ball query / factor Adam)
was in flight, from a rocprofv3 kernel trace of the default bench.  usage: overlap_inflation.py <trace dir>"""
import collections, csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]) for r in rows)
side = [e for e in ev if any(s in e[2] for s in ("fps_kernel", "ball_query", "adam_lowrank_kernel"))]
half = ev[len(ev) // 2][0]        # steady state only
stat = collections.defaultdict(lambda: [0, 0.0, 0, 0.0, 0.0, 0.0])
for s, e, n in ev:
    if s < half or any(k in n for k in ("fps_kernel", "ball_query", "adam_lowrank_kernel")):
        continue
    ov_f = sum(max(0, min(e, x[1]) - max(s, x[0])) for x in side if "fps" in x[2] or "ball" in x[2])
    ov_a = sum(max(0, min(e, x[1]) - max(s, x[0])) for x in side if "adam" in x[2])
    st = stat[n]
    d = (e - s) / 1e3
    if ov_f + ov_a < 0.1 * (e - s):
        st[0] += 1; st[1] += d
    else:
        st[2] += 1; st[3] += d; st[4] += ov_f / 1e3; st[5] += ov_a / 1e3
tot_alone = tot_over = 0
print(f"{'kernel':58s} alone n/us   overlapped n/us  (fps-ov us, adam-ov us)")
for n, st in sorted(stat.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
    a = st[1] / st[0] if st[0] else 0
    o = st[3] / st[2] if st[2] else 0
    if st[1] + st[3] > 200:
        print(f"{n[:58]:58s} {st[0]:4d} {a:7.1f}   {st[2]:4d} {o:7.1f}   ({st[4] / max(st[2], 1):.0f}, {st[5] / max(st[2], 1):.0f})")
