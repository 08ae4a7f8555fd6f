"""Run-to-run spread of the bf16 chain's weight-gradient error against the pre-rounded oracle (a steady value = deterministic kernels;
a spread beyond the atomics' noise = a hazard).  usage: python tools/err_probe.py"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from maskplanner_amd import sa_mlp
from oracle import torch_ref as T
import test_gpu_bf16 as tb
B, S, K, cin, widths = 2, 16, 128, 3, [64, 96, 128]
errs = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    convs, bns = tb._chain(cin, widths, seed=7 * cin + K)
    layers = tb._layers(convs, bns)
    convs.cuda(), bns.cuda().train()
    g = torch.Generator().manual_seed(K + 1)
    x = torch.randn(B, S, K, cin, generator=g) * 0.2
    gout = torch.randn(B, S, widths[-1], generator=g)
    out = sa_mlp.shared_mlp_max(x.cuda(), convs, bns, dtype="bf16")
    (out * gout.cuda()).sum().backward()
    st16 = 1 if (cin <= 4 and T.bf16_storage(1, widths, K)) else 0
    ref = T.shared_mlp_max(x.clone(), layers, True, bf16=True, store16=st16)
    (ref * gout).sum().backward()
    errs.append([round(tb.rel_l2(c.weight.grad.reshape(c.out_channels, -1), L["weight"].grad), 5) for c, L in zip(convs, layers)])
print(errs if len(errs) <= 4 else sorted(set(map(tuple, errs))))
