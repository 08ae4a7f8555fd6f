"""One-screen summary of a bench.py JSON line.  usage: bench_summary.py <file>"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("value", round(d["value"], 1), d["unit"], "| ms/step mean", round(d["ms_per_step"], 4), "median", round(d.get("step_ms_median", 0), 4), "min", round(d.get("step_ms_min", 0), 4))
for k in ("dropin_path", "streamed_inputs", "ucube"):
    if k in d:
        print(k, round(d[k]["value"], 1), "pc/s", round(d[k]["ms_per_step"], 3), "ms, median", round(d[k].get("step_ms_median") or 0, 3))
r = d.get("roofline", {})
print("roofline", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k in ("kernel", "bound", "achieved", "peak", "frac", "avg_us", "traffic")})
print("binding", r.get("binding"), "executed", {k: round(v, 4) for k, v in r.get("executed", {}).items() if isinstance(v, float)})
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"])
print("top kernels", dict(list(d.get("kernels_us_per_step", {}).items())[:14]))
