"""List the torch (ATen) operators of one training step that launch device kernels, grouped by the maskplanner_amd source
line that issued them -- to see where the small elementwise / fill / reduce launches come from."""
import collections
import os
import sys

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
ev = prof.events()
by = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    # only leaf ops (those that directly own kernels)
    where = "?"
    for fr in (e.stack or []):
        if "maskplanner_amd" in fr:
            where = fr.split("/")[-1][:70]
            break
    if where == "?" and e.stack:
        where = "[autograd/other] " + e.stack[0][-50:]
    dev_us = sum(k.duration for k in e.kernels)
    kn = ",".join(sorted({k.name[:24] for k in e.kernels if "Mem" in k.name}))
    key = (e.name + (" [" + kn + "]" if kn else ""), where)
    by[key][0] += len(e.kernels)
    by[key][1] += dev_us
tot = sum(v[0] for v in by.values())
print("kernels launched by ATen ops in one step:", tot)
for (name, where), (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{us:8.1f} us {n:4d}  {name:40s} {where}")
