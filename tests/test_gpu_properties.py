"""Size-independent properties of the HIP kernels at BASELINE.json's full sizes (B=32, N=5120; the loss shapes of
cuboids / shelves) -- checks that do not need the (slower) CPU oracle at that size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cloud():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from maskplanner_amd import synthetic as syn
    rng = np.random.default_rng(77)
    return torch.from_numpy(syn.point_cloud(rng, 32, 5120, "cuboid")).cuda()


def test_fps_properties_full_size(cloud):
    from maskplanner_amd import ops
    B, N, _ = cloud.shape
    start = torch.randint(0, N, (B,)).cuda()
    idx, new_xyz = ops.fps(cloud, 512, start, return_xyz=True)
    assert torch.equal(idx[:, 0], start)                                   # emits the start first (:80)
    assert (idx >= 0).all() and (idx < N).all()
    assert all(len(torch.unique(idx[b])) == 512 for b in range(B))          # distinct points are never re-selected
    assert torch.equal(new_xyz, torch.gather(cloud, 1, idx[:, :, None].expand(-1, -1, 3)))
    # greedy property: the (s+1)-th sample is a farthest point from the first s+1 samples (checked at a few s)
    for s in (1, 17, 300, 510):
        d = ((cloud[:, :, None, :] - new_xyz[:, None, : s + 1, :]) ** 2).sum(-1).min(-1)[0]   # [B,N]
        chosen = torch.gather(d, 1, idx[:, s + 1: s + 2])[:, 0]
        assert torch.allclose(chosen, d.max(1)[0], rtol=1e-6, atol=0)
    # determinism and prefix property: sampling fewer points yields a prefix
    assert torch.equal(ops.fps(cloud, 512, start), idx)
    assert torch.equal(ops.fps(cloud, 128, start), idx[:, :128])


def test_ball_query_properties_full_size(cloud):
    from maskplanner_amd import ops
    B, N, _ = cloud.shape
    idx_f, new_xyz = ops.fps(cloud, 512, torch.zeros(B, dtype=torch.long).cuda(), return_xyz=True)
    K, r = 32, 0.2
    g = ops.ball_query(r, K, cloud, new_xyz)                                # [B,512,K]
    assert (g >= 0).all() and (g < N).all()                                 # queries are cloud points: never empty
    sq = ops.square_distance(new_xyz, cloud)                                # the reference's expanded-form distances
    r2 = np.float32(r * r)
    inball = ~(sq > float(r2))
    picked = torch.gather(inball, 2, g)
    assert picked.all()                                                     # every returned index is inside the ball
    cnt = inball.sum(-1).clamp(max=K)                                       # [B,512]
    ar = torch.arange(K, device=g.device)[None, None]
    real = ar < cnt[..., None]
    # ascending order among the real hits, padding repeats the first hit
    assert ((g[..., 1:] > g[..., :-1]) | ~real[..., 1:]).all()
    assert ((g == g[..., :1]) | real).all()
    # "first K in index order": the number of in-ball points with index <= the last real hit equals the hit count
    last = torch.gather(g, 2, (cnt - 1)[..., None])
    below = (inball & (torch.arange(N, device=g.device)[None, None] <= last)).sum(-1)
    assert torch.equal(below, cnt)
    # the query itself is always a member
    assert (torch.gather(inball, 2, idx_f[..., None])).all()


@pytest.mark.parametrize("S,Sgt,Pgt", [(999, 985, 2959), (1266, 1148, 3448)])
def test_knn_chamfer_properties_full_size(S, Sgt, Pgt):
    from maskplanner_amd import ops
    from maskplanner_amd.pytorch3d_chamfer import chamfer_distance
    g = torch.Generator().manual_seed(S)
    x = torch.rand(32, S, 24, generator=g).cuda()
    y = torch.rand(32, Sgt, 24, generator=g).cuda()
    d, i = ops.knn(x, x, None, None, 2)
    assert (d[..., 0] == 0).all() and torch.equal(i[..., 0], torch.arange(S).cuda()[None].expand(32, -1))  # self match
    assert (d[..., 1] >= d[..., 0]).all()                                                                 # ascending
    d1, i1 = ops.knn(x, y, None, None, 1)
    bi = torch.arange(32).cuda()[:, None]
    assert torch.allclose(d1[..., 0], ((x - y[bi, i1[..., 0]]) ** 2).sum(-1), rtol=1e-5, atol=1e-6)         # index/value agree
    # padding invariance: -100 rows appended to y change nothing (pytorch3d_chamfer.py:138-149)
    ypad = torch.cat([y, torch.full((32, 40, 24), -100.0).cuda()], 1)
    a = chamfer_distance(x, y, asymmetric=True, point_reduction=None, batch_reduction=None)[0]
    b = chamfer_distance(x, ypad, padded=True, asymmetric=True, point_reduction=None, batch_reduction=None)[0]
    assert torch.equal(a, b)
    # symmetry: chamfer(x, y) == chamfer(y, x); zero on identical sets
    assert torch.allclose(chamfer_distance(x, y)[0], chamfer_distance(y, x)[0], rtol=1e-6)
    assert float(chamfer_distance(x, x)[0]) == 0.0
    # pose-cloud call of the loss (D=6) at full size: permutation of the references permutes the indices only
    p = torch.rand(32, 4 * S, 6, generator=g).cuda()
    q = torch.rand(32, Pgt, 6, generator=g).cuda()
    dq, iq = ops.knn(q, p, None, None, 1)
    perm = torch.randperm(4 * S, generator=g).cuda()
    dq2, iq2 = ops.knn(q, p[:, perm], None, None, 1)
    assert torch.equal(dq, dq2) and torch.equal(perm[iq2], iq)


def test_mask_match_is_a_valid_optimal_assignment_full_size():
    from scipy.optimize import linear_sum_assignment
    from maskplanner_amd import ops
    g = torch.Generator().manual_seed(5)
    B, M, S = 32, 41, 1266                                           # shelves_v2
    pred = (torch.randn(B, M, S, generator=g) * 2).cuda()
    ids = torch.randint(0, 41, (B, S), generator=g).float().cuda()
    match, uniq, nt, status, cost = ops.mask_match(pred, ids, return_cost=True)
    assert (status == 0).all()
    match, nt, cost = match.cpu().numpy(), nt.cpu().numpy(), cost.cpu().numpy()
    for b in range(B):
        k = int(nt[b])
        m = match[b]
        used = m[m >= 0]
        assert len(used) == min(M, k) and len(np.unique(used)) == len(used) and used.max() < k   # one-to-one
        r, c = linear_sum_assignment(cost[b, :, :k].astype(np.float64))
        got = cost[b, np.nonzero(m >= 0)[0], used].astype(np.float64).sum()
        assert abs(got - cost[b, r, c].astype(np.float64).sum()) <= 1e-9 * abs(got)              # optimal total cost
