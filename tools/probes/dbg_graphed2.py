import sys, torch
sys.path.insert(0, '.')
from maskplanner_amd import graphed, ops, pointnet2_cls_ssg as pc, pointnet2_utils as pu, synthetic
graphed.ENABLED = False
torch.manual_seed(3)
m = pc.maskplanner_model(synthetic.CATEGORIES["cuboids"], hidden_size=(256, 256)).cuda().eval()
g = torch.Generator().manual_seed(10)
x = (torch.rand(4, 1024, 3, generator=g) * 2 - 1).permute(0, 2, 1).cuda()
def rel(a, b): return float((a.double()-b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12))
def starts(): pu._fps_start_queue[:] = [[1,2,3,4], [5,6,7,8]]
with torch.no_grad():
    for _ in range(3):
        starts(); want = m.encode(x)
    # manual capture of encode with static starts
    xs = torch.empty(4, 1024, 3, device="cuda").permute(0, 2, 1); xs.copy_(x)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    st = []
    pu._capture_starts = st
    with torch.cuda.graph(gr):
        feat = m.encode(xs)
    pu._capture_starts = None
    print("starts recorded:", [(tuple(t.shape), n) for t, n in st])
    st[0][0].copy_(torch.tensor([1,2,3,4], device="cuda")); st[1][0].copy_(torch.tensor([5,6,7,8], device="cuda"))
    gr.replay(); torch.cuda.synchronize()
    print("encode: graph vs eager", rel(feat, want))
    # stage by stage, eager with the same starts
    pm = pu._points_major(xs)
    s1 = torch.tensor([1,2,3,4], device="cuda"); s2 = torch.tensor([5,6,7,8], device="cuda")
    f1, n1 = ops.fps(pm, 512, s1, return_xyz=True)
    g2 = torch.cuda.CUDAGraph()
    sx = torch.zeros(4, dtype=torch.long, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.graph(g2):
        f1g, n1g = ops.fps(pm, 512, sx, return_xyz=True)
        i1g = ops.ball_query(0.2, 32, pm, n1g)
    sx.copy_(s1); g2.replay(); torch.cuda.synchronize()
    i1 = ops.ball_query(0.2, 32, pm, n1)
    print("fps idx equal:", bool((f1 == f1g).all()), "ball idx equal:", bool((i1 == i1g).all()))
