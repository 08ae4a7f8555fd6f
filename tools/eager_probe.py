"""Wall-clock of eagerly launched steps interleaved with graph replays (the first eager step after recording pays for the
allocator blocks of its stream: bench.py runs one during set-up)."""
import os, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from maskplanner_amd.harness import TrainStep
ts = TrainStep("cuboids", B=32, N=5120)
for _ in range(8): ts.step()
torch.cuda.synchronize()
for i in range(6):
    t0 = time.perf_counter(); 
    if i % 2 == 0: ts.eager_step()
    else: ts.step()
    torch.cuda.synchronize(); print("eager" if i % 2 == 0 else "graph", round((time.perf_counter() - t0) * 1e3, 2), "ms", flush=True)
