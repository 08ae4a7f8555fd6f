"""Hunt for non-finite values in long graph-replay runs: step like bench.py does (graph replays, every 10th step eager),
synchronise after every step and stop at the first non-finite loss / parameter / optimizer moment."""
import sys
import torch
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskplanner_amd.harness import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
interleave = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sync_every = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ts = TrainStep("cuboids", B=32, N=5120)
names = [n for n, _ in ts.model.named_parameters()]
params = [p for _, p in ts.model.named_parameters()]
hist = []
for s in range(steps):
    loss = ts.eager_step() if (interleave and s % interleave == 0 and ts._graph is not None) else ts.step()
    if s % sync_every:
        continue
    flags = torch.stack([torch.isfinite(p).all() for p in params])
    gflags = [bool(torch.isfinite(p.grad).all()) if p.grad is not None else True for p in params]
    if not all(gflags):
        print("step", s, "non-finite GRAD in", [n for n, f in zip(names, gflags) if not f])
        for n, p in zip(names, params):
            if p.grad is not None:
                bad = (~torch.isfinite(p.grad)).nonzero()
                if len(bad):
                    print("  ", n, tuple(p.shape), "bad", len(bad), "first", bad[:3].tolist(), "vals", p.grad[~torch.isfinite(p.grad)][:4].tolist())
    ok = bool(flags.all()) and bool(torch.isfinite(loss)) and all(gflags)
    hist.append(float(loss))
    if not ok:
        print("step", s, "loss", float(loss), "last losses", hist[-6:])
        print("non-finite params:", [n for n, f in zip(names, flags.tolist()) if not f][:20])
        for n, p in zip(names, params):
            if p.grad is not None and not bool(torch.isfinite(p.grad).all()):
                print("non-finite grad:", n)
        for n, p in zip(names, params):
            bad = (~torch.isfinite(p)).nonzero()
            if len(bad):
                print(n, tuple(p.shape), "bad entries", len(bad), "first", bad[:4].tolist(), "cols", sorted(set(bad[:, 1].tolist()))[:12] if bad.shape[1] > 1 else "")
                st = ts.opt.state.get(p, {})
                for k, v in st.items():
                    if torch.is_tensor(v) and v.numel() == p.numel():
                        print("   state", k, "non-finite", int((~torch.isfinite(v)).sum()))
        for extra in range(3):
            print("next loss", float(ts.step()))
        break
else:
    print("clean", steps, "steps; loss first/last", hist[0], hist[-1], "max", max(hist))
