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
    if wz is not None and any(t in k for t in ("gemm", "knn", "fps", "ball", "pool", "group", "fused", "roles", "chunk", "adam", "dw_ci4", "first", "linear", "lean", "mask_match")):
        traffic[k.split("(")[0]] = dict(fetch_kib=fz, write_kib=wz, hbm_bytes=(2 * fz + wz) * 1024)
# the other BASELINE configs (tools/profile_configs.sh): one table per config key, looked up by bench.py before the default one
configs = {}
for d in sorted(glob.glob(f"{go}/{tag}_cfg_*_fetch")):
    key = os.path.basename(d)[len(tag) + 5:-6]
    def one(kind):
        fs = glob.glob(f"{go}/{tag}_cfg_{key}_{kind}/*/*_counter_collection.csv")
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(max(fs, key=os.path.getmtime))) if fs else []:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in agg.items()}
    cf, cw = one("fetch"), one("write")
    tab = {}
    for k, fz in cf.items():
        wz = cw.get(k)
        if wz is not None and any(t in k for t in ("gemm", "knn", "fps", "ball", "pool", "group", "fused", "roles", "chunk", "adam", "dw_ci4", "first", "linear", "lean", "mask_match", "lsap", "stream16", "pair", "head", "rc_stats")):
            tab[k.split("(")[0]] = dict(fetch_kib=fz, write_kib=wz, hbm_bytes=(2 * fz + wz) * 1024)
    configs[key] = tab
json.dump(dict(source=f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes ({tag}), bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch; "
                      "`kernels`: the default bench (cuboids, SSG, fp32); `configs`: the same two passes over the other BASELINE configs",
               kernels=traffic, configs=configs), open(f"{out}/{tag}_traffic.json", "w"), indent=1)
print("wrote", len(traffic), "traffic entries,", {k: len(v) for k, v in configs.items()})

# ---- MFMA utilisation of the hot grouped-MLP kernels (its own --pmc pass) ------------------------------------------------------
import glob as _g
mf = _g.glob(f"{go}/{tag}_mfma/*/*_counter_collection.csv")
if mf:
    rows = list(csv.DictReader(open(max(mf, key=os.path.getmtime))))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {short(r["Name"]): float(r["AverageNs"]) for r in stats}
    with open(f"{out}/{tag}_mfma_util.md", "w") as f:
        f.write(f"# {tag}: MFMA utilisation of the grouped-MLP kernels\n\n"
                "`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -- python3 bench.py --steps 3 "
                "--warmup 1 --no-cpu-baseline --no-side-legs` (counters only, own pass), per-launch averages.\n\n"
                "* `GRBM_GUI_ACTIVE` is summed over the 8 XCDs: busy shader cycles per XCD = GUI/8; held clock = GUI/8 / duration "
                "(duration from the separate `--kernel-trace --stats` run of the same command).\n"
                "* `SQ_VALU_MFMA_BUSY_CYCLES` is summed over the chip's 1024 SIMDs: **MFMA utilisation = BUSY / (1024 x GUI/8)**.\n"
                "* fp32 MFMA (`v_mfma_f32_32x32x2_f32`, 2048 MACs) holds a SIMD's matrix pipe for 64 cycles, `16x16x4` (1024 MACs) for 32: "
                "the fp32 peak of 157.3 TFLOP/s is 1024 SIMDs x 32 MAC/cycle x 2.4 GHz; at the held clock the peak scales down "
                "with it, so `util x held/2.4` is the fraction of the nominal peak the MFMA pipe was busy for.\n"
                "* the split-plane kernels (names ending `, true>` / `, 3>`) issue `v_mfma_f32_32x32x16_bf16` (16384 MACs, 32 busy cycles) and "
                "`v_mfma_f32_16x16x32_bf16` (8192 MACs, 16 cycles): six of them per fp32 product tile, so BUSY = 6 x algorithmic MACs / 512 per cycle "
                "-- e.g. `bwd_fused_kernel<3, 256, 128>`: 2 x 8.6e9 MACs x 6 / 512 = 2.013e8 SIMD-cycles, the counter's value to four digits -- and "
                "their MFMA utilisation is a fraction of the bf16 dense peak (2.5 PFLOP/s), not of the fp32 one.\n\n"
                "| kernel | launches | avg us | GUI_ACTIVE/8 (cycles) | held clock GHz | MFMA_BUSY (SIMD-cycles) | MFMA util | util x clk/2.4 |\n|---|---|---|---|---|---|---|---|\n")
        names = sorted(agg, key=lambda n: -dur.get(n, 0.0) * len(agg[n].get("GRBM_GUI_ACTIVE", [])))
        for n in names:
            if not any(t in n for t in ("fused", "roles", "chunk", "gemm", "bwd_first", "dw_ci4", "rc_stats", "bf16")):
                continue
            c = agg[n]
            gui = sum(c["GRBM_GUI_ACTIVE"]) / max(len(c["GRBM_GUI_ACTIVE"]), 1) / 8.0
            busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / max(len(c["SQ_VALU_MFMA_BUSY_CYCLES"]), 1)
            us = dur.get(n, 0.0) / 1e3
            clk = gui / (us * 1e3) if us else 0.0
            util = busy / (1024.0 * gui) if gui else 0.0
            f.write(f"| `{n[:90]}` | {len(c['GRBM_GUI_ACTIVE'])} | {us:.1f} | {gui:.0f} | {clk:.2f} | {busy:.3e} | {100 * util:.1f} % | {100 * util * clk / 2.4:.1f} % |\n")
    print("wrote mfma util")
