"""Run each set-abstraction level's fused MLP (forward + backward) a few times: target for rocprofv3 --kernel-trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd import sa_mlp

torch.manual_seed(0)
shapes = [(32, 512, 32, 3, [64, 64, 128]), (32, 128, 64, 131, [128, 128, 256]), (32, 1, 128, 259, [256, 512, 1024])]
for B, S, K, C0, mlp in shapes:
    convs, bns = torch.nn.ModuleList(), torch.nn.ModuleList()
    last = C0
    for c in mlp:
        convs.append(torch.nn.Conv2d(last, c, 1)); bns.append(torch.nn.BatchNorm2d(c)); last = c
    convs, bns = convs.cuda(), bns.cuda()
    x = torch.randn(B, S, K, C0).cuda().requires_grad_(C0 != 3)
    g = torch.randn(B, S, mlp[-1]).cuda()
    for it in range(3):
        y = sa_mlp.shared_mlp_max(x, convs, bns, layout="xyz_first" if C0 == 3 or C0 == 259 else "feats_first")
        (y * g).sum().backward()
torch.cuda.synchronize()
print("done")
