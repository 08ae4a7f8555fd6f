"""Worker of tests/test_gpu_variants.py: one set-abstraction level (forward + backward) under the environment it was started with; saves the
output and every gradient.  The library reads its switches once per process, hence a process per variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd import sa_mlp

shape, out = sys.argv[1], sys.argv[2]
B, S, K, C0, mlp = {"sa2": (8, 128, 64, 131, [128, 128, 256]), "sa3": (16, 1, 128, 259, [256, 512, 1024]),
                     "mid256": (8, 64, 32, 131, [128, 256, 128])}[shape]      # mid256: an INTERIOR 128 -> 256 layer (the non-pooled form of the 256-output kernels)
torch.manual_seed(5)
convs, bns = torch.nn.ModuleList(), torch.nn.ModuleList()
last = C0
for c in mlp:
    convs.append(torch.nn.Conv2d(last, c, 1)); bns.append(torch.nn.BatchNorm2d(c)); last = c
convs, bns = convs.cuda(), bns.cuda()
x = torch.randn(B, S, K, C0).cuda().requires_grad_(True)
g = torch.randn(B, S, mlp[-1]).cuda()
y = sa_mlp.shared_mlp_max(x, convs, bns, layout="feats_first")
(y * g).sum().backward()
res = dict(y=y.detach().cpu(), gx=x.grad.cpu())
for i, (c, b) in enumerate(zip(convs, bns)):
    res[f"dw{i}"] = c.weight.grad.cpu()
    res[f"dg{i}"] = b.weight.grad.cpu()
    res[f"db{i}"] = b.bias.grad.cpu()
    res[f"rm{i}"] = b.running_mean.cpu()
torch.save(res, out)
