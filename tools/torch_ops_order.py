"""One eager training step: every ATen / autograd op that launches device kernels, in launch order, with the maskplanner_amd frame
that issued it -- the map from the small launches of the step back to source lines."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskplanner_amd.harness import TrainStep  # noqa: E402

ts = TrainStep("cuboids", B=32, N=5120, graph=False)
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True,
                            experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    ts.step()
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    if any(c.kernels for c in (e.cpu_children or [])):
        continue      # not a leaf
    where = "?"
    for fr in (e.stack or []):
        if "maskplanner_amd" in fr:
            where = fr.split("/")[-1][:60]
            break
    rows.append((e.time_range.start, e.name, where, [k.name[:40] for k in e.kernels], sum(k.duration for k in e.kernels)))
rows.sort()
for t, n, w, ks, us in rows:
    print(f"{us:7.1f} us  {n[:34]:34s} {w:60s} {ks[0] if ks else ''}{' +' + str(len(ks) - 1) if len(ks) > 1 else ''}")
