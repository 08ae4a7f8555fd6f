"""Read a rocprofv3 kernel trace (csv) and report, for every fps_kernel<256,...> launch, which kernels ran concurrently."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
fps = [e for e in ev if "fps_kernel<256" in e[2]]
print("launches", len(ev), "fps", len(fps))
for s, e, n, q in fps[-4:]:
    over = [(x[2][:50], x[3], max(0, min(e, x[1]) - max(s, x[0])) / 1e3) for x in ev if x[1] > s and x[0] < e and x[2] != n]
    tot = sum(o[2] for o in over)
    print(f"fps {q} {(e - s) / 1e3:.0f} us; concurrent kernels {len(over)}, overlapped kernel time {tot:.0f} us")
    for o in over[:12]:
        print("    ", o)
# step time from fps start to fps start
st = [e[0] for e in fps]
print("fps-to-fps us:", [round((b - a) / 1e3) for a, b in zip(st[:-1], st[1:])][-8:])
# idle gaps inside the last full step
if len(fps) >= 3:
    s0, s1 = fps[-3][0], fps[-2][0]
    step = [x for x in ev if x[0] >= s0 and x[0] < s1]
    busy_end, gaps = s0, []
    for x in step:
        if x[0] > busy_end + 3000:
            gaps.append(((x[0] - busy_end) / 1e3, x[2][:60]))
        busy_end = max(busy_end, x[1])
    print("step", (s1 - s0) / 1e3, "us, kernels", len(step), "gaps > 3 us:", len(gaps), "total", round(sum(g[0] for g in gaps)), "us")
    for g in sorted(gaps, reverse=True)[:12]:
        print("   gap %.1f us before %s" % g)
    print("sum of kernel durations", round(sum(x[1] - x[0] for x in step) / 1e3), "us")
