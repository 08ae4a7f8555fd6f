"""Where the host is at a step boundary: per-call host time of the pieces of TrainStep._step (sampling hand-over + launches, graph
replays, factor Adam launches) over N replayed steps without synchronising, next to the device step time.
usage: boundary_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd.harness import TrainStep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ts = TrainStep("cuboids", B=32, N=5120, device="cuda:0")
while ts.use_graph and ts._graph is None:
    ts.step()
for _ in range(5):
    ts.step()
torch.cuda.synchronize()
acc = {}
def wrap(obj, name, key=None):
    f = getattr(obj, name)
    key = key or name
    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc.setdefault(key, []).append(time.perf_counter() - t)
        return r
    setattr(obj, name, g)
wrap(ts, "_launch_sampling")
wrap(ts, "_launch_factor_adam")
wrap(ts._graph, "replay", "graphA.replay")
if ts._graph_b is not None:
    wrap(ts._graph_b, "replay", "graphB.replay")
marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
t0 = time.perf_counter()
for i in range(n):
    marks[i].record()
    t = time.perf_counter()
    ts.step()
    host.append(time.perf_counter() - t)
marks[-1].record()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
dev = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(n))
print(f"device step ms: median {dev[n // 2]:.3f} min {dev[0]:.3f}; wall {1e3 * (t2 - t0) / n:.3f} ms/step; host enqueue loop {1e3 * (t1 - t0) / n:.3f} ms/step")
host.sort()
print(f"host per step(): median {1e3 * host[n // 2]:.3f} ms, max {1e3 * host[-1]:.3f}")
for k, v in acc.items():
    v.sort()
    print(f"  {k:22s} calls/step {len(v) / n:.1f}  median {1e6 * v[len(v) // 2]:.0f} us  max {1e6 * v[-1]:.0f} us")
