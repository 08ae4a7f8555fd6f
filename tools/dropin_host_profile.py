"""cProfile of the drop-in loop's Python side (harness.DropInLoop: the reference's loop body on the drop-in modules), by total and by
cumulative time: where the host spends the step it is bound by.  usage: dropin_host_profile.py"""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskplanner_amd.harness import DropInLoop
loop = DropInLoop("cuboids", B=32, N=5120)
for _ in range(24):
    loop.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    loop.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(40)
