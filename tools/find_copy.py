"""Find device-to-device memcpy operations inside one training step (they become memcpy NODES in a recorded step) and print
the shapes they move -- used to track down `dists[..., 0]`, whose backward is a fill + a memcpy."""
import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from maskplanner_amd.harness import TrainStep
ts = TrainStep("cuboids", B=32, N=5120, graph=False)
for _ in range(3): ts.step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    ts.step(); torch.cuda.synchronize()
seen = 0
for e in prof.events():
    if e.name == "aten::copy_" and e.kernels and any("Mem" in k.name for k in e.kernels):
        print("---", [k.name for k in e.kernels], e.input_shapes, "thread", e.thread, "seq", e.sequence_nr)
        for fr in (e.stack or [])[:14]: print("    ", fr[-110:])
        seen += 1
        if seen >= 3: break
