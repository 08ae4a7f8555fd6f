"""After tools/refresh_profiles.sh <tag> has run on the GPU box and gpurun merged gpurun_out/ back: write everything under profiles/<tag>_*
(kernel stats, traffic, MFMA utilisation, overlap, SQ counters, the bench lines) and print the headline numbers."""
import json, os, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
lines_only = "--lines-only" in sys.argv
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
if not lines_only:
    subprocess.run([sys.executable, "tools/make_profile_summary.py", tag], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([sys.executable, "tools/make_overlap_summary.py", tag, f"gpurun_out/{tag}_overlap"], check=True, stdout=subprocess.DEVNULL)


def line(f):
    for l in open(f):
        if l.startswith("{"):
            return l


for name, f in (("bench_line", "bench_full"), ("config3_windows", "windows"), ("config4_shelves_1gpu", "shelves"), ("config5_containers_msg_bf16", "c5_bf16"),
                ("config5_containers_msg_f32", "c5_f32"), ("fp32_mfma_kernels", "fp32mfma")):
    l = line(f"gpurun_out/{tag}_{f}.log")
    open(f"profiles/{tag}_{name}.json", "w").write(l)
    d = json.loads(l)
    r = d.get("roofline", {})
    print(name, round(d["value"], 1), round(d["ms_per_step"], 3), d.get("step_ms_median") and round(d["step_ms_median"], 3), r.get("kernel"), r.get("bound"),
          round(r.get("achieved", 0), 1), round(r.get("frac", 0), 3), r.get("avg_us") and round(r["avg_us"], 1))
    if name == "bench_line":
        for k in ("dropin_path", "streamed_inputs", "ucube"):
            print("   ", k, round(d[k]["value"], 1), round(d[k]["ms_per_step"], 3))
        print("    executed", {a: (round(b, 3) if isinstance(b, float) else b) for a, b in (r.get("executed") or {}).items() if a != "what"})
        print("    traffic", r.get("traffic"), "cpu", round(d["cpu_baseline"]["value"], 2), d["cpu_baseline"]["cores"])
if lines_only:
    sys.exit(0)
hdr = open(f"profiles/{tag}_sq_counters.md").read().split("```")[0]
body = open(f"gpurun_out/{tag}_pmc_hot.txt").read()
body = body[body.index("kernel "):]
open(f"profiles/{tag}_sq_counters.md", "w").write(hdr + "```\n" + body.strip() + "\n```\n")
print(open(f"profiles/{tag}_overlap.md").read()[-900:])
