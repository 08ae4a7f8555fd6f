import sys, torch
sys.path.insert(0, '.')
from maskplanner_amd import graphed, ops, sa_mlp, pointnet2_cls_ssg as pc, pointnet2_utils as pu, synthetic
graphed.ENABLED = False
pc.SAMPLE_AHEAD = False
torch.manual_seed(3)
m = pc.maskplanner_model(synthetic.CATEGORIES["cuboids"], hidden_size=(256, 256)).cuda().eval()
g = torch.Generator().manual_seed(10)
x = (torch.rand(4, 1024, 3, generator=g) * 2 - 1).permute(0, 2, 1).cuda()
def rel(a, b): return float((a.double()-b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12))
def starts(): pu._fps_start_queue[:] = [[1,2,3,4], [5,6,7,8]]
def stages(xin):
    sa_mlp.prepermute([(m.sa1.mlp_convs[0], "xyz_first"), (m.sa2.mlp_convs[0], "feats_first"), (m.sa3.mlp_convs[0], "feats_first")])
    a, b = m.sa1(xin, None)
    c, d = m.sa2(a, b)
    e, f = m.sa3(c, d)
    return [a, b, c, d, f]
with torch.no_grad():
    for _ in range(3):
        starts(); want = stages(x)
    starts(); want2 = stages(x)
    print("eager vs eager:", [rel(a, b) for a, b in zip(want2, want)])
    xs = torch.empty(4, 1024, 3, device="cuda").permute(0, 2, 1); xs.copy_(x)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    st = []
    pu._capture_starts = st
    with torch.cuda.graph(gr):
        got = stages(xs)
    pu._capture_starts = None
    st[0][0].copy_(torch.tensor([1,2,3,4], device="cuda")); st[1][0].copy_(torch.tensor([5,6,7,8], device="cuda"))
    gr.replay(); torch.cuda.synchronize()
    print("graph vs eager [l1_xyz, l1_points, l2_xyz, l2_points, l3_points]:", [rel(a, b) for a, b in zip(got, want)])
    gr.replay(); torch.cuda.synchronize()
    print("second replay:", [rel(a, b) for a, b in zip(got, want)])
