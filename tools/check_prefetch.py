import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd.harness import TrainStep
from maskplanner_amd import pointnet2_utils as pu
ts = TrainStep("cuboids", B=8, N=5120, hidden_size=(256, 256), prefetch_sampling=True)
ts.model.eval()
with torch.no_grad():
    l0 = float(ts.forward_loss())   # inline sampling, then queues a prefetch
    assert pu.has_prefetched(ts.batch["point_cloud"], 512, 0.2, 32)
    l1 = float(ts.forward_loss())   # consumes the prefetched plan
    ts.prefetch = False
    pu._prefetched.clear()
    l2 = float(ts.forward_loss())
print("inline", l0, "prefetched", l1, "inline again", l2)
assert l0 == l1 == l2
# timing: steps with and without prefetch
import time
for pf in (False, True):
    ts = TrainStep("cuboids", B=32, N=5120, prefetch_sampling=pf)
    for _ in range(5): ts.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ts.step()
    torch.cuda.synchronize(); print("prefetch", pf, (time.perf_counter() - t0) / 20 * 1e3, "ms/step")
