"""Experiment (not product): the drop-in model's forward + backward replayed from hipGraphs (torch.cuda.make_graphed_callables) inside the
reference's loop body -- how much of the drop-in loop's host time that would remove.  FPS start indices are fixed device tensors here
(the product would feed the CPU-drawn ones in as graph inputs); the sample-ahead side stream is off."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskplanner_amd.harness import DropInLoop
from maskplanner_amd import pointnet2_utils as pu, pointnet2_cls_ssg as ssg

def run(loop, n=40):
    for _ in range(8): loop.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): loop.step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

base = DropInLoop("cuboids", B=32, N=5120)
print(f"drop-in loop as shipped: {run(base):.2f} ms/step")
ssg.SAMPLE_AHEAD = False
loop = DropInLoop("cuboids", B=32, N=5120)
model = loop.model.train()
B, N = 32, 5120
s1 = torch.randint(0, N, (B,), device="cuda"); s2 = torch.randint(0, model.sa1.npoint, (B,), device="cuda")
orig_draw = pu._draw_fps_start
queue = []
pu._draw_fps_start = lambda b, n, dev: (s1 if n == N else s2)
x = loop.host_batches[0]["point_cloud"].permute(0, 2, 1).to("cuda", dtype=torch.float).contiguous()
class Tensors(torch.nn.Module):       # (make_graphed_callables wants tensor outputs only: the model returns None for absent heads)
    def __init__(self, m):
        super().__init__(); self.m = m; self.mask = None
    def forward(self, x):
        out = self.m(x); self.mask = [o is not None for o in out]
        return tuple(o for o in out if o is not None)
wrapped = Tensors(model)
def graphed(x):
    outs = iter(g_(x))
    return tuple(next(outs) if keep else None for keep in wrapped.mask)
try:
    g_ = torch.cuda.make_graphed_callables(wrapped, (x,), num_warmup_iters=3)
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300]); sys.exit(0)
loop.model_call = graphed
import types
def step(self):
    import numpy as np
    data = self.host_batches[self._i % len(self.host_batches)]; self._i += 1
    self.model.train(); self.model.zero_grad()
    pc = data["point_cloud"].permute(0, 2, 1).to(self.device, dtype=torch.float); traj = data["traj"].to(self.device, dtype=torch.float)
    traj_pred, pm, ms, sl = graphed(pc)
    loss, ll = self.loss_handler.compute(y_pred=traj_pred, y=traj, pred_stroke_masks=pm, mask_scores=ms, seg_logits=sl,
                                         stroke_ids=data["stroke_ids"], traj_as_pc=data["traj_as_pc"])
    loss.backward(); self.opt.step(); v = loss.item(); self.model.zero_grad(); return v
loop.step = types.MethodType(step, loop)
print(f"model forward + backward from graphs: {run(loop):.2f} ms/step; last loss {loop.step():.3f}")
