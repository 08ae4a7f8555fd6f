"""Functional dry run of the multi-rank path with several ranks on ONE GPU (gloo backend).  Launch with torchrun."""
import faulthandler, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(90, exit=True)
import torch, torch.distributed as dist
from maskplanner_amd import dp
rank, local, world = dp.init_from_env()
def log(*a):
    print(f"[rank {rank}]", *a, flush=True)
log("init done, backend", dist.get_backend())
from maskplanner_amd.harness import TrainStep
ts = TrainStep("cuboids", B=4, N=1024, hidden_size=(128, 128), rank=rank)
log("model built; buckets", len(ts.reducer.buckets), "dense MB", ts.reducer.grad_bytes() / 1e6)
for i in range(3):
    loss = ts.step()
    torch.cuda.synchronize()
    log("step", i, "loss", float(loss))
# replicas must stay identical
flat = torch.cat([p.detach().reshape(-1) for p in ts.model.parameters()])
ref = flat.clone()
dist.broadcast(ref, src=0)
log("max param difference vs rank 0:", float((flat - ref).abs().max()))
dist.barrier()
dist.destroy_process_group()
log("done")
