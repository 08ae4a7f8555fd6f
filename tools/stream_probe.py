"""Where the host time of a streamed step goes (TrainStep(stream_batches=K))."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd.harness import TrainStep
ts = TrainStep("cuboids", B=32, N=5120, stream_batches=4)
for _ in range(8): ts.step()
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
for _ in range(30): ts.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
pr.disable()
print("host ms/step", (t1 - t0) / 30 * 1e3, "wall ms/step", (t2 - t0) / 30 * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
