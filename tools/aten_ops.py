"""Which torch operators still launch kernels inside one training step: an eager step under torch.profiler, top-level aten ops with
input shapes and the Python line that issued them.  usage: aten_ops.py [c5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from maskplanner_amd.harness import TrainStep
if len(sys.argv) > 1 and sys.argv[1] == "c5":
    ts = TrainStep("containers", B=32, N=10240, encoder="msg", mlp_dtype="bf16")
else:
    ts = TrainStep("cuboids", B=32, N=5120)
for _ in range(6):
    ts.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    ts.eager_step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.cpu_parent is None or
       (e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.cpu_parent is not None and not e.cpu_parent.name.startswith("aten::"))]
evs.sort(key=lambda e: e.time_range.start)
for e in evs:
    ks = [k.name[:40] for k in e.kernels] if hasattr(e, "kernels") else []
    dev = e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
    if dev <= 0:
        continue
    st = [s for s in (e.stack or []) if "maskplanner_amd" in s or "harness" in s][:2]
    print(f"{e.name:32s} {str(e.input_shapes)[:70]:70s} dev {dev:7.1f}us  {' | '.join(s.split('maskplanner_amd/')[-1][:60] for s in st)}")
