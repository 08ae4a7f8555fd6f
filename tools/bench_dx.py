"""Stand-alone time of the two head input-gradient kernels (csrc/linear_dx.hip, csrc/adam_lowrank.hip) and torch's GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd import _lib, ops
lib = _lib.load()
for B, O, I in [(32, 11988, 1024), (32, 5994, 1024), (32, 15192, 1024)]:
    g = torch.randn(B, O, device="cuda"); W = torch.randn(O, I, device="cuda") / 32
    gx = torch.empty(B, I, device="cuda")
    ws = torch.empty((lib.mp_linear_dx_skinny_workspace_bytes(B, O, I),), dtype=torch.uint8, device="cuda")
    fns = {"mfma": lambda: ops._run("linear_dx_mfma", g, lib.mp_linear_dx_mfma_f32, g.data_ptr(), W.data_ptr(), B, O, I, gx.data_ptr()),
           "skinny": lambda: ops._run("linear_dx_skinny", g, lib.mp_linear_dx_skinny_f32, g.data_ptr(), W.data_ptr(), B, O, I, gx.data_ptr(), ws.data_ptr(), ws.numel()),
           "torch": lambda: torch.matmul(g, W, out=gx)}
    big = torch.empty(64 << 20, device="cuda")     # flush the caches between launches
    for name, fn in fns.items():
        ts = []
        for _ in range(12):
            big.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        print(f"O={O}: {name:7s} {ts[len(ts)//2]:7.1f} us  ({O * I * 4 / ts[len(ts)//2] / 1e6:.2f} TB/s)")
