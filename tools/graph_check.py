"""hipGraph replay of the training step vs the eager path.

1. Same seeds, same batch: loss sequences of an eager run, a second eager run (run-to-run spread: the dW GEMMs use fp32
   atomics) and a graph run; host enqueue time and time-to-device-done per step for both modes.
2. Two identical trainers in capturable mode, one replaying its recorded step and one launching kernel by kernel: after the
   same number of steps their gradients and parameters must agree to within that atomics noise.
"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskplanner_amd.harness import TrainStep


def run(graph, steps=30):
    torch.manual_seed(0)
    torch.cuda.manual_seed(0)
    ts = TrainStep("cuboids", B=32, N=5120, graph=graph)
    out = [float(ts.step()) for _ in range(steps)]   # graph mode returns the SAME tensor every step: read it per step
    t0 = time.perf_counter()
    for _ in range(50):
        ts.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return out, ts._graph is not None, 1e3 * (t1 - t0) / 50, 1e3 * (t2 - t0) / 50


def pair(steps_after=2, record=True):
    def make():
        torch.manual_seed(0)
        torch.cuda.manual_seed(0)
        return TrainStep("cuboids", B=32, N=5120, graph=True)
    a, b = make(), make()
    b.GRAPH_AFTER = 10 ** 9                      # same capturable optimizers, never records
    if not record:
        a.GRAPH_AFTER = 10 ** 9                  # baseline: two eager trainers
    for m in (a, b):
        for p in m.model.modules():
            if isinstance(p, torch.nn.Dropout):
                p.p = 0.0                        # the two trainers share one Philox stream: keep them comparable
    for _ in range(TrainStep.GRAPH_AFTER + steps_after):
        a.step()
        b.step()
    torch.cuda.synchronize()
    assert (a._graph is not None) == record and b._graph is None
    # global relative differences (per-tensor ratios are meaningless for tensors whose gradient is analytically zero -- a conv
    # bias in front of a train-mode BatchNorm -- or whose value is a few noise-driven Adam steps away from zero)
    num_g = den_g = num_p = den_p = 0.0
    for (n, pa), (_, pb) in zip(a.model.named_parameters(), b.model.named_parameters()):
        num_p += float((pa - pb).double().square().sum())
        den_p += float(pb.double().square().sum())
        if pa.grad is not None and pb.grad is not None:
            num_g += float((pa.grad - pb.grad).double().square().sum())
            den_g += float(pb.grad.double().square().sum())
    return (num_g / den_g) ** 0.5, (num_p / den_p) ** 0.5


if __name__ == "__main__":
    e, _, eh, ed = run(False)
    e2, _, _, _ = run(False)
    g, used, gh, gd = run(True)
    print("graph recorded:", used)
    print("eager2 losses:", [round(x, 3) for x in e2[:6]], "...", round(e2[-1], 3), "(run-to-run spread of the eager path)")
    print("eager  losses:", [round(x, 3) for x in e[:6]], "...", round(e[-1], 3), f"| host {eh:.2f} ms/step, device-done {ed:.2f} ms/step")
    print("graph  losses:", [round(x, 3) for x in g[:6]], "...", round(g[-1], 3), f"| host {gh:.2f} ms/step, device-done {gd:.2f} ms/step")
    bg, bp = pair(record=False)
    print(f"eager  vs eager after {TrainStep.GRAPH_AFTER + 2} steps: relative gradient difference {bg:.2e}, parameter difference {bp:.2e} (noise baseline)")
    wg, wp = pair()
    print(f"replay vs eager after {TrainStep.GRAPH_AFTER + 2} steps: relative gradient difference {wg:.2e}, parameter difference {wp:.2e} (all tensors together)")
