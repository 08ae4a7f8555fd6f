"""[r6] Is the loss a replayed step returns ordered on the caller's current stream?  One trainer on a side stream, the loss cloned after every
step WITHOUT a host synchronisation, read at the end -- next to the same with float() per step."""
import sys, torch
sys.path.insert(0, '.')
from maskplanner_amd.harness import TrainStep
def run(stream, mode, n=8):
    with torch.cuda.stream(stream):
        ts = TrainStep("cuboids", B=4, N=1024, seed=11, graph=True)
        got = []
        for _ in range(n):
            l = ts.step()
            got.append(float(l) if mode == "float" else (l.clone() if mode == "clone" else l.detach() + 0))
    torch.cuda.synchronize()
    return [round(float(x), 2) for x in got]
for mode in ("float", "clone", "add"):
    print("default", mode, run(torch.cuda.default_stream(), mode))
    print("side   ", mode, run(torch.cuda.Stream(), mode))
