"""Host-side timeline of the replayed training step: how long each part of TrainStep.step() takes to ISSUE, vs the device time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd.harness import TrainStep

ts = TrainStep("cuboids", B=32, N=5120)
for _ in range(12):
    ts.step()
torch.cuda.synchronize()
assert ts._graph is not None
marks = {}
orig = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        marks.setdefault(label, []).append(time.perf_counter() - t0)
        return r
    setattr(obj, name, g)
wrap(ts, "_launch_sampling", "launch_sampling (next plan on the side stream)")
wrap(ts._graph, "replay", "graph A replay")
wrap(ts._graph_b, "replay", "graph B replay")
wrap(ts, "_launch_factor_adam", "factor Adam launches")
t0 = time.perf_counter()
N = 200
for _ in range(N):
    ts.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time per step {(t1 - t0) / N * 1e3:.3f} ms; wall per step {(t2 - t0) / N * 1e3:.3f} ms")
for k, v in marks.items():
    v = sorted(v)
    print(f"  {k}: median {v[len(v) // 2] * 1e6:.0f} us, p90 {v[int(len(v) * 0.9)] * 1e6:.0f} us")
