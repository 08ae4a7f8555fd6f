"""[r6] The drop-in loop on a pool of ragged-width host batches: per-step wall clock and host-synchronised phase times over one pass of the pool,
for pools of 4 and 32 batches (what the pool's size itself costs: pageable H2D from 32 different host tensors)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from maskplanner_amd.harness import DropInLoop
from maskplanner_amd import graphed

for nb in (4, 32):
    loop = DropInLoop("cuboids", B=32, N=5120, n_batches=nb)
    for _ in range(24):
        loop.step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(64):
        t0 = time.perf_counter()
        loop.step()
        ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    s = sorted(ts)
    print(f"pool {nb}: mean {np.mean(ts):.2f} median {s[len(s) // 2]:.2f} min {s[0]:.2f} max {s[-1]:.2f} ms", "| steps over 8 ms:", [round(t, 1) for t in ts if t > 8], flush=True)
    print("   ", graphed.loss_stats(loop.loss_handler))
    m = loop.model
    acc = {}
    for i in range(2 * nb):
        data = loop.host_batches[i % nb]
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
        if i >= nb:
            for k, v in dict(h2d=b - a, forward=c - b, loss=d - c, backward=e - d, adam=f - e).items():
                acc.setdefault(k, []).append(v * 1e3)
    print("    phases (median / max ms):", {k: (round(sorted(v)[len(v) // 2], 2), round(max(v), 2)) for k, v in acc.items()}, flush=True)
    del loop
    torch.cuda.empty_cache()
