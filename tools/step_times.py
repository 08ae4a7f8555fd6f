"""Per-step device times (HIP events on the main stream) of harness.TrainStep: sequence and quantiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maskplanner_amd.harness import TrainStep
ov = len(sys.argv) > 1 and sys.argv[1] == "1"
ts = TrainStep("cuboids", B=32, N=5120, overlap_sampling=ov)
for _ in range(10): ts.step()
torch.cuda.synchronize()
n = 100
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    ts.step(); ev[i + 1].record()
torch.cuda.synchronize()
t = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(n)])
print("overlap", ov, "mean %.3f median %.3f min %.3f p90 %.3f" % (t.mean(), np.median(t), t.min(), np.quantile(t, 0.9)))
print(np.round(t[:40], 2).tolist())
