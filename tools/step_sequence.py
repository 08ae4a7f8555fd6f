"""One replayed step of the default bench, kernel by kernel in start order: offset, duration, gap to the previous end on the chain.
usage: step_sequence.py <trace dir> [step index from the end, default 3]"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(csv.DictReader(open(f)))
def nm(r):
    return r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r), r.get("Stream_Id", "?"), r.get("Queue_Id", "?")) for r in rows)
starts = [e[0] for e in ev if "rc_stats_kernel" in e[2]]
s0, s1 = starts[-back - 1], starts[-back]
ks = [e for e in ev if s0 <= e[0] < s1]
side_names = ("fps_kernel", "ball_query", "adam_lowrank_kernel")
busy = s0
small = 0.0
nsmall = 0
for s, e, n, st, q in ks:
    side = any(x in n for x in side_names)
    gap = (s - busy) / 1e3 if not side else 0.0
    print(f"{(s - s0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {'S' if side else ' '} gap {gap:5.1f}  {n[:100]}")
    if not side:
        busy = max(busy, e)
        if e - s < 8000:
            small += (e - s) / 1e3; nsmall += 1
print("span", (s1 - s0) / 1e3, "kernels", len(ks), "chain kernels < 8 us:", nsmall, "sum", round(small), "us")
