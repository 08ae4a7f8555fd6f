import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd.harness import TrainStep
for pf in (False, True, False, True):
    ts = TrainStep("cuboids", B=32, N=5120, prefetch_sampling=pf)
    for _ in range(5): ts.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): ts.step()
    torch.cuda.synchronize(); print("prefetch", pf, round((time.perf_counter() - t0) / 30 * 1e3, 3), "ms/step")
