import sys, subprocess
if len(sys.argv) == 1:
    for v in ("test1", "notwin", "noevalprobe", "notwin_noevalprobe", "ones", "noseed"):
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True)
        print(v, "rc", r.returncode, [l for l in r.stdout.splitlines() if l.startswith("ok")], flush=True)
    sys.exit(0)
v = sys.argv[1]
import torch
sys.path.insert(0, '.')
from maskplanner_amd import graphed, pointnet2_cls_ssg as pc, synthetic
def model():
    torch.manual_seed(3)
    return pc.maskplanner_model(synthetic.CATEGORIES["cuboids"], hidden_size=(256, 256)).cuda().train()
def clouds(seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(4, 1024, 3, generator=g) * 2 - 1).permute(0, 2, 1).cuda()
eager, rec = model(), model()
xs = [clouds(10 + i) for i in range(6)]
if "noevalprobe" in v:
    shapes = [(4, 1500, 6), (4, 6, 500), (4, 6)]
    graphed.ENABLED = False
    probe = [o for o in eager(xs[0]) if o is not None]
    graphed.ENABLED = True
    probe = [o.detach() for o in probe]
else:
    with torch.no_grad():
        probe = [o for o in eager.eval()(xs[0]) if o is not None]
    eager.train()
torch.manual_seed(0)
gouts = [torch.ones_like(o) if v == "ones" else torch.randn_like(o) for o in probe]
for i, x in enumerate(xs):
    for m, on in ((eager, False), (rec, True)):
        if "notwin" in v and not on: continue
        graphed.ENABLED = on
        m.zero_grad()
        if v != "noseed": torch.manual_seed(100 + i)
        outs = m(x)
        keep = [o for o in outs if o is not None]
        torch.autograd.backward(keep, gouts)
        graphed.ENABLED = True
print("ok", [r.graph_r is not None for r in rec._graph_runners.values()])
