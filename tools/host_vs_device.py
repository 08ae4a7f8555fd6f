"""How far ahead of the GPU does the host run?  Enqueue K steps without synchronising and compare the time the Python loop
took (host) with the time until the device finished them (device)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskplanner_amd.harness import TrainStep
ts = TrainStep("cuboids", B=32, N=5120)
for _ in range(10):
    ts.step()
torch.cuda.synchronize()
for K in (20, 50):
    t0 = time.perf_counter()
    for _ in range(K):
        ts.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K={K}: host enqueue {1e3 * (t1 - t0) / K:.2f} ms/step, until device done {1e3 * (t2 - t0) / K:.2f} ms/step")
