"""Turn gpurun_out/<tag>_{bench,fetch,write} rocprofv3 CSVs into profiles/<tag>_*.md / .csv (committed evidence)."""
import collections, csv, glob, json, os, sys


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out")
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)

def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "")

stats = list(csv.DictReader(open(newest(f"{go}/{tag}_bench/*/*_kernel_stats.csv"))))
with open(f"{out}/{tag}_bench_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in stats:
        w.writerow([short(r["Name"])[:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

def pmc(kind):
    rows = list(csv.DictReader(open(newest(f"{go}/{tag}_{kind}/*/*_counter_collection.csv"))))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fetch, write = pmc("fetch"), pmc("write")
line = json.loads(next(l for l in open(f"{go}/{tag}_bench.log") if l.startswith("{")))
with open(f"{out}/{tag}_bench_kernel_stats.md", "w") as f:
    f.write(f"# {tag}: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline\n\n")
    f.write("bench line of the same run (under the profiler):\n\n```json\n" + json.dumps(line) + "\n```\n\n")
    f.write("Separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes (bench.py --steps 3): per-launch averages in KiB as rocprofv3 reports them.  "
            "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so HBM read bytes ~= 2 x FETCH_SIZE x 1024; "
            "WRITE_SIZE x 1024 is exact for 16-B stores (other widths uncalibrated).\n\n")
    f.write("| kernel | calls | avg us | total % | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | est. HBM MB/launch (2xF + W) |\n|---|---|---|---|---|---|---|\n")
    for r in stats[:40]:
        n = short(r["Name"])
        fz = next((v for k, v in fetch.items() if k.startswith(n[:60])), None)
        wz = next((v for k, v in write.items() if k.startswith(n[:60])), None)
        est = "" if fz is None or wz is None else f"{(2 * fz + wz) * 1024 / 1e6:.1f}"
        f.write(f"| `{n[:100]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} | "
                f"{'' if fz is None else round(fz)} | {'' if wz is None else round(wz)} | {est} |\n")
traffic = {}
for k, fz in fetch.items():
    wz = write.get(k)
    if wz is not None and any(t in k for t in ("gemm", "knn", "fps", "ball", "pool", "group", "fused", "chunk", "adam", "dw_ci4")):
        traffic[k.split("(")[0]] = dict(fetch_kib=fz, write_kib=wz, hbm_bytes=(2 * fz + wz) * 1024)
json.dump(dict(source=f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes ({tag}), bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch",
               kernels=traffic), open(f"{out}/{tag}_traffic.json", "w"), indent=1)
print("wrote", len(traffic), "traffic entries")
