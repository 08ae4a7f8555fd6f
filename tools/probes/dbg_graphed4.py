import sys, subprocess, os
if len(sys.argv) == 1:
    for v in ("allouts", "allouts_hidden1024", "allouts_seed3", "allouts_other", "allouts_B8", "allouts_N2048"):
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True)
        print(v, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1][:200], flush=True)
    sys.exit(0)
v = sys.argv[1]
import torch
sys.path.insert(0, '.')
from maskplanner_amd import graphed, pointnet2_cls_ssg as pc, synthetic
torch.manual_seed(3 if "seed3" in v else 5)
hs = (1024, 1024) if "hidden1024" in v else (256, 256)
m = pc.maskplanner_model(synthetic.CATEGORIES["cuboids"], hidden_size=hs).cuda().train()
other = pc.maskplanner_model(synthetic.CATEGORIES["cuboids"], hidden_size=hs).cuda().train() if "other" in v else None
BB = 8 if "B8" in v else 4
NN = 2048 if "N2048" in v else 1024
def clouds(seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(BB, NN, 3, generator=g) * 2 - 1).permute(0, 2, 1).cuda()
x = clouds(40)
for i in range(graphed.WARM + 2):
    if "nozero" not in v: m.zero_grad()
    if "seed" in v: torch.manual_seed(i)
    if "newx" in v: x = clouds(40 + i)
    if other is not None:
        graphed.ENABLED = False
        oo = other(x); oo[0].sum().backward(); del oo
        graphed.ENABLED = True
    out = m(x)
    if v.startswith("sum0"): out[0].sum().backward()
    elif v == "sum_all": sum(o.sum() for o in out if o is not None).backward()
    else: torch.autograd.backward([o for o in out if o is not None], [torch.ones_like(o) for o in out if o is not None])
print("ok", [r.graph_r is not None for r in m._graph_runners.values()])
