"""[r5] The unchanged loop on recorded graphs for a few thousand steps: loss trend, no growth of allocated memory, no fallback."""
import sys, time, torch
sys.path.insert(0, '.')
from maskplanner_amd.harness import DropInLoop
loop = DropInLoop("cuboids", B=32, N=5120)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
vals, mem = [], []
t0 = time.perf_counter()
for i in range(n):
    vals.append(loop.step())
    if i % 250 == 249:
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated() >> 20)
        print(i + 1, "steps: mean loss of the last 250 %.2f" % (sum(vals[-250:]) / 250), "allocated MiB", mem[-1], "ms/step %.3f" % ((time.perf_counter() - t0) / (i + 1) * 1e3), flush=True)
rs = loop.model._graph_runners
print("model runners:", [(k[0], r.graph_r is not None, r.failed, r.calls) for k, r in rs.items()])
print("loss runners:", [(r.graph_lb is not None, r.failed, r.calls) for r in loop.loss_handler._graph_runners.values()])
assert all(v == v for v in vals) and mem[-1] <= mem[1] + 64
