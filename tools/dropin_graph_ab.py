"""[r5] The reference's loop body on the drop-in modules with the model's forward / backward replayed from recorded graphs (maskplanner_amd/graphed.py)
against the same loop launched op by op: ms per step (median), and that both give the same losses from the same seed.
usage: python tools/dropin_graph_ab.py [B] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd import graphed
from maskplanner_amd.harness import DropInLoop

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5120


def run(on, steps=40, adam=None):
    graphed.ENABLED = on
    torch.manual_seed(7)
    loop = DropInLoop("cuboids", B=B, N=N, adam_kwargs=adam)
    losses = [loop.step() for _ in range(8)]
    per = []
    for _ in range(steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        losses.append(loop.step())
        torch.cuda.synchronize(); per.append((time.perf_counter() - t0) * 1e3)
    per.sort()
    rec = sum(1 for r in loop.model.__dict__.get("_graph_runners", {}).values() if r.graph_r is not None)
    return per[len(per) // 2], sum(per) / len(per), losses, rec


for adam in (None, {"fused": True}):
    for rnd in range(2):
        for on in (False, True):
            med, mean, losses, rec = run(on, adam=adam)
            print(f"Adam {adam or 'foreach'}  graphs {'on ' if on else 'off'}: median {med:.2f} ms  mean {mean:.2f} ms  recorded shapes {rec}  "
                  f"loss[0, 7, 20, 47] = {losses[0]:.4f} {losses[7]:.4f} {losses[20]:.4f} {losses[47]:.4f}", flush=True)
