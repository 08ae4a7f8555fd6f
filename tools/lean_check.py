"""The pooled layer without its stored activation (csrc/sa_lean.hip) against the stored-Z form and an fp64 torch evaluation, per
parameter, at the two sampled levels of BASELINE configs[1]; then forward / backward times of both forms.
    python tools/lean_check.py [--ucube-like]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd import sa_mlp

torch.manual_seed(0)
shapes = [(32, 512, 32, 3, [64, 64, 128]), (32, 128, 64, 131, [128, 128, 256])]


def ref64(x, convs, bns):
    h = x.double()
    B, S, K, C = h.shape
    h = h.reshape(-1, C)
    for conv, bn in zip(convs, bns):
        w = conv.weight.double().view(conv.out_channels, -1)
        z = h @ w.t() + conv.bias.double()
        mean, var = z.mean(0), z.var(0, unbiased=False)
        h = torch.relu((z - mean) / torch.sqrt(var + bn.eps) * bn.weight.double() + bn.bias.double())
    return h.view(B, S, K, -1).max(2).values


for B, S, K, C0, mlp in shapes:
    convs, bns = torch.nn.ModuleList(), torch.nn.ModuleList()
    last = C0
    for c in mlp:
        convs.append(torch.nn.Conv2d(last, c, 1)); bns.append(torch.nn.BatchNorm2d(c)); last = c
    convs, bns = convs.cuda(), bns.cuda()
    for bn in bns:
        torch.nn.init.uniform_(bn.weight, -1.0, 1.0)
        torch.nn.init.normal_(bn.bias, 0.0, 0.3)
    x = torch.randn(B, S, K, C0).cuda()
    if "--dup" in sys.argv:      # ball-query padding: most members of a group are copies of its first
        x[:, :, 5:] = x[:, :, :1]
    if C0 != 3:
        x = torch.cat([torch.relu(x[..., :-3]), 0.2 * x[..., -3:]], -1)
    x.requires_grad_(C0 != 3)
    g = torch.randn(B, S, mlp[-1]).cuda()
    layout = "xyz_first" if C0 == 3 else "feats_first"
    names = [n for n, _ in list(convs.named_parameters()) + list(bns.named_parameters())]
    params = list(convs.parameters()) + list(bns.parameters())
    res = {}
    for lean in (False, True):
        sa_mlp.LEAN_LAST = lean
        for p in params:
            p.grad = None
        x.grad = None
        y = sa_mlp.shared_mlp_max(x, convs, bns, layout=layout)
        (y * g).sum().backward()
        res[lean] = (y.detach().clone(), [p.grad.clone() for p in params], None if x.grad is None else x.grad.clone())
    xr = x.detach() if C0 == 3 else torch.cat([x.detach()[..., -3:], x.detach()[..., :-3]], -1)
    xr = xr.double().requires_grad_(True)
    for p in params:
        p.grad = None
    y64 = ref64(xr, convs, bns)
    (y64 * g.double()).sum().backward()
    g64 = [p.grad.clone() for p in params]
    print(f"level C0={C0} mlp={mlp} K={K}")
    for lean in (False, True):
        y, gr, gx = res[lean]
        print(f"  {'lean  ' if lean else 'stored'}: fwd err {float((y.double() - y64).abs().max() / y64.abs().max()):.2e}", end="")
        if gx is not None:
            g64x = xr.grad[..., 3:]
            print(f"  dX rel-L2 {float((gx[..., :-3].double() - g64x).norm() / g64x.norm()):.2e}", end="")
        print()
        for n, a, b in zip(names, gr, g64):
            if b.abs().max() > 1e-6:
                print(f"      {n:12s} {float((a.double() - b.double()).norm() / b.double().norm()):.2e}", end="")
        print()
    ys, grs, gxs = res[False]
    yl, grl, gxl = res[True]
    print("  lean vs stored: out", float((ys - yl).abs().max()), " grads", " ".join(f"{float((a - b).norm() / b.norm().clamp_min(1e-30)):.1e}" for a, b in zip(grl, grs)),
          "" if gxs is None else f" dX {float((gxl - gxs).norm() / gxs.norm()):.1e}")
    for p in params:
        p.grad = None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for lean in (False, True):
        sa_mlp.LEAN_LAST = lean
        for _ in range(3):
            y = sa_mlp.shared_mlp_max(x, convs, bns, layout=layout); (y * g).sum().backward()
        torch.cuda.synchronize()
        tf = tb = 0.0
        for _ in range(10):
            ev[0].record(); y = sa_mlp.shared_mlp_max(x, convs, bns, layout=layout); ev[1].record(); (y * g).sum().backward(); ev[2].record()
            torch.cuda.synchronize()
            tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
        print(f"  {'lean  ' if lean else 'stored'}: fwd {tf * 100:.0f} us, bwd {tb * 100:.0f} us")
