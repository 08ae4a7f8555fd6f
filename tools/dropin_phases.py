"""Where the drop-in loop's step goes: host-synchronised phase times (H2D, model forward, loss, backward, optimizer).
[r3] 4.79 ms/step: h2d 0.14, forward 1.22, loss 0.57, backward 2.1, torch.optim.Adam 0.86 ms.  (A hipGraph replay of the model's forward +
backward through torch.cuda.make_graphed_callables was tried: forward unchanged -- it is not host-bound --, backward +0.4 ms for the
copy of the static gradients into param.grad: 5.06 ms/step.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maskplanner_amd.harness import DropInLoop

from maskplanner_amd import graphed
for kw, on in (({}, False), ({}, True), ({"fused": True}, False), ({"fused": True}, True)):   # fused=True: a one-word change of train_maskplanner.py:159
    graphed.ENABLED = on          # [r5] the model's forward / backward replayed from recorded graphs (maskplanner_amd/graphed.py)
    loop = DropInLoop("cuboids", B=32, N=5120, adam_kwargs=kw)
    print("torch.optim.Adam kwargs:", kw, "| model graphs:", "on" if on else "off")
    for _ in range(24):      # (four host batches in rotation: every shape's recordings are behind us)
        loop.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        loop.step()
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step")
    # phases (synchronised: sums to more than the pipelined step)
    data = loop.host_batches[0]
    m = loop.model
    ph = {}
    for _ in range(5):
        m.zero_grad()
        torch.cuda.synchronize(); a = time.perf_counter()
        pc = data["point_cloud"].permute(0, 2, 1).to("cuda", dtype=torch.float); traj = data["traj"].to("cuda", dtype=torch.float)
        torch.cuda.synchronize(); b = time.perf_counter()
        out = m(pc)
        torch.cuda.synchronize(); c = time.perf_counter()
        loss, ll = loop.loss_handler.compute(y_pred=out[0], y=traj, pred_stroke_masks=out[1], mask_scores=out[2], seg_logits=out[3],
                                             stroke_ids=data["stroke_ids"], traj_as_pc=data["traj_as_pc"])
        torch.cuda.synchronize(); d = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize(); e = time.perf_counter()
        loop.opt.step()
        torch.cuda.synchronize(); f = time.perf_counter()
        ph = dict(h2d=b - a, forward=c - b, loss=d - c, backward=e - d, adam=f - e)
    print("   ", {k: round(v * 1e3, 2) for k, v in ph.items()})
