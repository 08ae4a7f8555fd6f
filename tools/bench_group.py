import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maskplanner_amd import ops
B, N, S, K, D = 32, 512, 128, 64, 128
xyz = torch.rand(B, N, 3).cuda(); feats = torch.randn(B, N, D).cuda().requires_grad_(True)
new_xyz = xyz[:, :S].contiguous(); idx = torch.randint(0, N, (B, S, K)).cuda()
g = torch.randn(B, S, K, 132).cuda()
for det in (False, True):
    ops.DETERMINISTIC = det
    for _ in range(3):
        feats.grad = None; ops.group(xyz, feats, new_xyz, idx, xyz_last=True, pad_to=4).backward(g)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = ops.group(xyz, feats, new_xyz, idx, xyz_last=True, pad_to=4)
    a.record()
    for _ in range(10):
        feats.grad = None; out.backward(g, retain_graph=True)
    b.record(); torch.cuda.synchronize()
    print("deterministic", det, a.elapsed_time(b) / 10 * 1e3, "us per group_bwd")
