import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from maskplanner_amd import sa_mlp
from test_gpu_modules import _torch_shared_mlp_max
B, S, K, C0, mlp = 8, 512, 32, 3, [64, 64, 128]
for variant in ("neg_gamma_dups", "neg_gamma_nodups", "pos_gamma_dups"):
    torch.manual_seed(B * S + C0)
    convs, bns = torch.nn.ModuleList(), torch.nn.ModuleList()
    last = C0
    for c in mlp:
        convs.append(torch.nn.Conv2d(last, c, 1)); bns.append(torch.nn.BatchNorm2d(c)); last = c
    convs, bns = convs.cuda(), bns.cuda()
    with torch.no_grad():
        for bn in bns:
            bn.weight.uniform_(0.4, 1.5)
            if variant.startswith("neg"):
                bn.weight.mul_(torch.where(torch.rand_like(bn.weight) < 0.25, -1.0, 1.0))
            bn.bias.uniform_(-0.3, 0.3)
    x = torch.randn(B, S, K, C0).cuda()
    if variant.endswith("_dups"):
        x[:, :, K // 2:] = x[:, :, :1]
    gout = torch.randn(B, S, mlp[-1]).cuda()
    res = []
    for fn in (sa_mlp.shared_mlp_max, _torch_shared_mlp_max):
        xi = x.clone().requires_grad_(True)
        for p in list(convs.parameters()) + list(bns.parameters()):
            p.grad = None
        y = fn(xi, convs, bns)
        (y * gout).sum().backward()
        res.append((xi.grad.clone(), y.detach().clone(), [p.grad.clone() for p in convs.parameters()], [p.grad.clone() for p in bns.parameters()]))
    d = (res[0][0] - res[1][0]).abs()
    bad = (d > 1e-3).nonzero()
    print(variant, "out err", float((res[0][1] - res[1][1]).abs().max()), "gx max err", float(d.max()), "n bad", len(bad), "first bad", bad[:5].tolist())
    print("   per-position max", [round(float(d[:, :, k].max()), 4) for k in range(K)])
    print("   conv grad errs", [round(float((a - b).abs().max()), 6) for a, b in zip(res[0][2], res[1][2])], " bn grad errs", [round(float((a - b).abs().max()), 6) for a, b in zip(res[0][3], res[1][3])])
