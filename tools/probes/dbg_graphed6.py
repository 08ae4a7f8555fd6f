import sys, torch
sys.path.insert(0, '.')
from maskplanner_amd import graphed, pointnet2_cls_ssg as pc, synthetic
torch.manual_seed(5)
m = pc.maskplanner_model(synthetic.CATEGORIES["cuboids"], hidden_size=(256, 256)).cuda().train()
g = torch.Generator().manual_seed(40)
x = (torch.rand(4, 1024, 3, generator=g) * 2 - 1).permute(0, 2, 1).cuda()
for i in range(graphed.WARM + 1):
    m.zero_grad(); m(x)[0].sum().backward()
r = next(iter(m._graph_runners.values()))
print("recorded", r.graph_r is not None)
name = "fc3.weight"
p = dict(m.named_parameters())[name]
def run(seed):
    torch.manual_seed(seed); o = m(x)[0]; print("  graphed node:", type(o.grad_fn).__name__); o.sum().backward()
m.zero_grad(); run(1); g1 = p.grad.clone(); print("after 1: is static", any(p.grad is g for _, g in r.grads), float(g1.norm()))
run(2); both = p.grad.clone(); print("after 2: is static", any(p.grad is g for _, g in r.grads), float(both.norm()))
m.zero_grad(); run(2); g2 = p.grad.clone(); print("G2", float(g2.norm()), "G1+G2", float((g1 + g2).norm()), "diff", float((both - g1 - g2).norm()))
m.zero_grad(); run(1); g1b = p.grad.clone(); print("G1 again diff", float((g1b - g1).norm()), float(g1.norm()))
graphed.ENABLED = False
m.zero_grad(); run(1); e1 = p.grad.clone(); print("eager G1 diff", float((e1 - g1).norm()))
graphed.ENABLED = True
def grads(): return {n: q.grad.clone() for n, q in m.named_parameters() if q.grad is not None}
m.zero_grad(); run(1); a = grads()
run(2); both = grads()
m.zero_grad(); run(2); b = grads()
gmax = max(float(v.abs().max()) for v in both.values())
rows = []
for n in both:
    w = a[n] + b[n]
    rows.append((float((both[n] - w).abs().max()) / max(float(w.abs().max()), 1e-3 * gmax), n, float(w.abs().max()), float(a[n].abs().max()), float(b[n].abs().max()), float(both[n].abs().max())))
rows.sort(reverse=True)
for r_ in rows[:6]: print(r_)
print("----")
m.zero_grad(); run(1)
ps = dict(m.named_parameters())
stat = {id(p_): g_ for p_, g_ in r.grads}
for n in ("fc3.bias", "sa1.mlp_bns.1.weight", "sa3.mlp_convs.2.bias", "fc3.weight"):
    q = ps[n]; g_ = stat[id(q)]
    print(n, "grad is static:", q.grad is g_, "same ptr:", q.grad.data_ptr() == g_.data_ptr(), "contig", g_.is_contiguous(), "shape", tuple(g_.shape), "storage_off", g_.storage_offset(), "base", g_._base is not None, q.grad.flatten()[:4].tolist())
ptrs = {}
for p_, g_ in r.grads:
    ptrs.setdefault(g_.data_ptr(), []).append(tuple(g_.shape))
print("aliased static grads:", [v for v in ptrs.values() if len(v) > 1][:5])
