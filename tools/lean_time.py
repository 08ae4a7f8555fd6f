"""Backward time of the two sampled levels with the lean pooled layer (per-kernel experiments: MP_LEAN_SKIP bit mask)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd import sa_mlp
torch.manual_seed(0)
for B, S, K, C0, mlp in [(32, 512, 32, 3, [64, 64, 128]), (32, 128, 64, 131, [128, 128, 256])]:
    convs, bns = torch.nn.ModuleList(), torch.nn.ModuleList()
    last = C0
    for c in mlp:
        convs.append(torch.nn.Conv2d(last, c, 1)); bns.append(torch.nn.BatchNorm2d(c)); last = c
    convs, bns = convs.cuda(), bns.cuda()
    x = torch.randn(B, S, K, C0).cuda()
    g = torch.randn(B, S, mlp[-1]).cuda()
    layout = "xyz_first" if C0 == 3 else "feats_first"
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(3):
        y = sa_mlp.shared_mlp_max(x, convs, bns, layout=layout); (y * g).sum().backward()
    torch.cuda.synchronize()
    tf = tb = 0.0
    for _ in range(10):
        ev[0].record(); y = sa_mlp.shared_mlp_max(x, convs, bns, layout=layout); ev[1].record(); (y * g).sum().backward(); ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    print(f"skip={os.environ.get('MP_LEAN_SKIP', '0')} lean={os.environ.get('MP_LEAN_LAST', '1')} C0={C0}: fwd {tf * 100:.0f} us, bwd {tb * 100:.0f} us")
