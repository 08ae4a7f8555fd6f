"""Screened (matrix-core) vs direct K = 1 nearest neighbours: identical outputs, and the time of each, on the three searches of the
training step (B = 32): GT points -> predicted poses (D = 6), predicted segments <-> GT segments (D = 24)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd import _lib, ops

lib = _lib.load()


def run(p1, p2, l1, l2, screened):
    B, P1, D = p1.shape
    d = torch.empty(B, P1, device="cuda")
    i = torch.empty(B, P1, dtype=torch.int64, device="cuda")
    nb = lib.mp_knn1_workspace_bytes(B, p2.shape[1], D)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    ops._run("knn1", p1, lib.mp_knn1_f32, p1.data_ptr(), p2.data_ptr(), None if l1 is None else l1.data_ptr(), None if l2 is None else l2.data_ptr(),
             B, P1, p2.shape[1], D, d.data_ptr(), i.data_ptr(), int(screened), ws.data_ptr(), ws.numel())
    return d, i


torch.manual_seed(0)
cases = [("gt points -> poses", 32, 3000, 3996, 6), ("segments -> gt", 32, 999, 700, 24), ("gt -> segments", 32, 700, 999, 24), ("xyz", 8, 1000, 5120, 3)]
for name, B, P1, P2, D in cases:
    p1 = torch.rand(B, P1, D, device="cuda")
    p2 = torch.rand(B, P2, D, device="cuda")
    p2[:, 5] = p2[:, 3]                      # duplicates: ties go to the first index
    p1[:, 7] = p2[:, 11]                     # zero distances
    l1 = torch.randint(P1 // 2, P1 + 1, (B,), device="cuda")
    l2 = torch.randint(P2 // 2, P2 + 1, (B,), device="cuda")
    d0, i0 = run(p1, p2, l1, l2, 0)
    d1, i1 = run(p1, p2, l1, l2, 1)
    same = bool(torch.equal(d0, d1) and torch.equal(i0, i1))
    ts = []
    for scr in (0, 1):
        for _ in range(3):
            run(p1, p2, l1, l2, scr)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(p1, p2, l1, l2, scr)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{name:22s} B={B} P1={P1} P2={P2} D={D}: identical={same}  direct {ts[0]:.1f} us  screened {ts[1]:.1f} us" +
          ("" if same else f"  mismatches: idx {int((i0 != i1).sum())} dist {int((d0 != d1).sum())}"))
