"""profiles/<tag>_overlap.md from a rocprofv3 kernel trace of the DEFAULT bench (two replayed graphs + eagerly launched sampling
and head optimizer): per replayed step, the time span, the kernel time on the step's stream and on the other streams, idle gaps.
usage: make_overlap_summary.py <tag> <trace dir>"""
import collections, csv, glob, os, statistics, sys
tag, d = sys.argv[1], sys.argv[2]
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "0")) for r in rows)
starts = [e[0] for e in ev if "rc_stats_kernel" in e[2]]          # first library kernel of graph A
steps = []
for s0, s1 in zip(starts[:-1], starts[1:]):
    ks = [e for e in ev if s0 <= e[0] < s1]
    steps.append((s1 - s0, ks))
steps = steps[-14:]
med = statistics.median(t for t, _ in steps)
replayed = [(t, ks) for t, ks in steps if t < 1.08 * med]              # drop the eagerly launched profiled steps
side_names = ("fps_kernel", "ball_query", "adam_lowrank_kernel")
out = [f"# {tag}: concurrency inside a replayed step (rocprofv3 --kernel-trace -- python3 bench.py --steps 20 --warmup 8 --no-cpu-baseline)\n",
       "Two recorded graphs (encoder forward | heads, loss, backward, dense Adam) on the step's stream; the NEXT batch's sampling plan (FPS + ball query of both levels) and the PREVIOUS step's factor Adam are launched eagerly on their own streams.\n",
       "| step | span us | kernel time, step's chain us | kernel time, other streams us | of which FPS / ball query / factor Adam | idle gaps > 2 us on the chain |", "|---|---|---|---|---|---|"]
for n, (t, ks) in enumerate(replayed[-6:]):
    side = [k for k in ks if any(s in k[2] for s in side_names) and not ("fps_kernel<64" in k[2] and False)]
    main = [k for k in ks if k not in side]
    busy, gaps = main[0][0], 0.0
    for k in main:
        if k[0] > busy + 2000:
            gaps += (k[0] - busy) / 1e3
        busy = max(busy, k[1])
    by = collections.Counter()
    for k in side:
        by["FPS" if "fps" in k[2] else ("ball query" if "ball" in k[2] else "factor Adam")] += (k[1] - k[0]) / 1e3
    out.append(f"| {n} | {t / 1e3:.0f} | {sum(k[1] - k[0] for k in main) / 1e3:.0f} | {sum(k[1] - k[0] for k in side) / 1e3:.0f} | "
               f"{by['FPS']:.0f} / {by['ball query']:.0f} / {by['factor Adam']:.0f} | {gaps:.0f} |")
out.append(f"\nmedian span of the last {len(steps)} steps under the profiler: {med / 1e3:.0f} us "
           f"({len(steps) - len(replayed)} eagerly launched profiled steps excluded from the table).\n")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
open(os.path.join(root, "profiles", f"{tag}_overlap.md"), "w").write("\n".join(out))
print("\n".join(out))
