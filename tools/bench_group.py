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
# forward gather (group_rows128_kernel for D = 128)
with torch.no_grad():
    for _ in range(3):
        ops.group(xyz, feats, new_xyz, idx, xyz_last=True, pad_to=4)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        out = ops.group(xyz, feats, new_xyz, idx, xyz_last=True, pad_to=4)
    b.record(); torch.cuda.synchronize()
    print("group forward", a.elapsed_time(b) / 20 * 1e3, "us")
    ref = torch.cat([ops.index_points(feats, idx), ops.index_points(xyz, idx) - new_xyz[:, :, None, :], torch.zeros(B, S, K, 1, device="cuda")], -1)
    print("equal to index_points form:", bool(torch.equal(out, ref)))
