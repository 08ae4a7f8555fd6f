import sys, time, torch
sys.path.insert(0, '.')
from maskplanner_amd.harness import TrainStep
ts = TrainStep("cuboids", B=32, N=5120)
while ts.use_graph and ts._graph is None: ts.step()
for _ in range(5): ts.step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
t0 = time.perf_counter()
for i in range(20):
    ev[i].record(); ts.step()
ev[20].record(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("wall per step %.4f ms" % (dt / 20 * 1e3), "device per step:", [round(ev[i].elapsed_time(ev[i+1]), 3) for i in range(20)])
print("first record -> last record %.4f ms/step" % (ev[0].elapsed_time(ev[20]) / 20))
