"""SyncBN under hipGraph replay, on the real RCCL backend with ONE forced rank (a gpurun box has one GPU): the step's per-layer all-reduces
(the library's exchange hook, sync_bn.Exchange) are recorded into the two graphs; the replayed trainer is compared with the same trainer
launched kernel by kernel.  Prints one JSON line.  usage: syncbn_graph_probe.py [port] [B] [N]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ["MASKPLANNER_SYNCBN_GRAPH"] = "1"     # (opt-in since r5: harness.TrainStep)
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29733")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 5120
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1)
import maskplanner_amd.dp as dp
dp.FORCE_COLLECTIVES = True
from maskplanner_amd.harness import TrainStep
out = {}
for mode, graph in (("eager", False), ("graph", None)):
    ts = TrainStep("cuboids", B=B, N=N, sync_bn=True, graph=graph)
    ts.model.dropout.p = 0.0
    losses = [float(ts.step()) for _ in range(12)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ts.step()
    torch.cuda.synchronize()
    out[mode] = {"recorded": ts._graph is not None, "sync_bn": bool(ts.sync_bn), "losses": losses, "ms": (time.perf_counter() - t0) / 20 * 1e3,
                 "finite": bool(all(torch.isfinite(p).all() for p in ts.model.parameters()))}
print(json.dumps(out))
dist.destroy_process_group()
