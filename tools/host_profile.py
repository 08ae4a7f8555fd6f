"""cProfile of the Python side of one training step (host enqueue cost, no synchronisation inside)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskplanner_amd.harness import TrainStep
ts = TrainStep("cuboids", B=32, N=5120)
for _ in range(10):
    ts.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    ts.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
