"""HIP kernels (through the C ABI) against the CPU oracle and the reference-generated golden vectors.

Bar: bit-exact for every index output (FPS, ball query, kNN indices, LAP assignment, padded lengths);
fp32 values within 1e-5 (north_star tolerance) -- kNN distances are compared bit-for-bit because the kernel
and the oracle share one rounding sequence.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from maskplanner_amd import ops as O
    return O


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _cases(g, suffix):
    return sorted({k[: -len(suffix)] for k in g.files if k.endswith(suffix)})


# ------------------------------------------------------------------------------------------------ FPS
def test_fps_golden_bit_exact(golden, ops):
    g = golden("g1_fps")
    for k in _cases(g, "_idx"):
        want = g[k + "_idx"]
        got, new_xyz = ops.fps(dev(g[k + "_xyz"]), want.shape[1], dev(g[k + "_start"]), return_xyz=True)
        assert got.dtype == torch.int64
        assert np.array_equal(got.cpu().numpy(), want), k
        xyz = g[k + "_xyz"]
        sel = np.take_along_axis(xyz, want[:, :, None].repeat(3, 2), axis=1)
        assert np.array_equal(new_xyz.cpu().numpy(), sel), k


@pytest.mark.parametrize("B,N,S,dist", [(32, 5120, 512, "cuboid"), (32, 5120, 512, "ucube"), (32, 512, 128, "cuboid"),
                                        (3, 10240, 512, "cuboid"), (2, 777, 100, "ucube"), (5, 64, 64, "ucube"),
                                        (2, 1, 1, "ucube"), (2, 2000, 300, "cuboid"), (1, 13312, 64, "ucube")])
def test_fps_vs_oracle(oracle, ops, B, N, S, dist):
    from maskplanner_amd import synthetic as syn
    rng = np.random.default_rng(N * 7 + S)
    xyz = syn.point_cloud(rng, B, N, dist)
    start = rng.integers(0, N, size=B)
    want = oracle.fps(xyz, S, start)
    got = ops.fps(dev(xyz), S, dev(start))
    assert np.array_equal(got.cpu().numpy(), want)


def test_fps_all_coincident(oracle, ops):
    xyz = np.zeros((2, 300, 3), np.float32) + 0.25
    start = np.array([7, 299])
    assert np.array_equal(ops.fps(dev(xyz), 40, dev(start)).cpu().numpy(), oracle.fps(xyz, 40, start))


# ------------------------------------------------------------------------------------------------ ball query
def test_ball_query_golden_bit_exact(golden, ops):
    g = golden("g2_bq")
    for k in _cases(g, "_idx"):
        got = ops.ball_query(float(g[k + "_radius"]), int(g[k + "_K"]), dev(g[k + "_xyz"]), dev(g[k + "_new_xyz"]))
        assert np.array_equal(got.cpu().numpy(), g[k + "_idx"].astype(np.int64)), k


def test_square_distance_golden_bit_exact(golden, ops):
    g = golden("g2_sqd")
    for t in "ab":
        got = ops.square_distance(dev(g[t + "_src"]), dev(g[t + "_dst"])).cpu().numpy()
        assert np.array_equal(got.view(np.int32), g[t + "_out"].view(np.int32))


@pytest.mark.parametrize("B,N,S,r,K,dist", [(32, 5120, 512, 0.2, 32, "cuboid"), (32, 5120, 512, 0.2, 32, "ucube"),
                                            (32, 512, 128, 0.4, 64, "cuboid"), (4, 10240, 512, 0.1, 16, "cuboid"),
                                            (4, 10240, 512, 0.4, 128, "cuboid"), (3, 777, 100, 0.35, 48, "ucube"),
                                            (2, 300, 7, 5.0, 400, "ucube"), (2, 100, 9, 1e-4, 8, "ucube")])
def test_ball_query_vs_oracle(oracle, ops, B, N, S, r, K, dist):
    from maskplanner_amd import synthetic as syn
    rng = np.random.default_rng(N + S + K)
    xyz = syn.point_cloud(rng, B, N, dist)
    fidx = oracle.fps(xyz, S, rng.integers(0, N, size=B))
    new_xyz = oracle.index_points(xyz, fidx)
    want = oracle.ball_query(r, K, xyz, new_xyz)
    got = ops.ball_query(r, K, dev(xyz), dev(new_xyz))
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("B,N,S,radii,Ks,dist", [(4, 10240, 512, [0.1, 0.2, 0.4], [16, 32, 128], "cuboid"), (4, 10240, 512, [0.1, 0.2, 0.4], [16, 32, 128], "ucube"),
                                                 (32, 512, 128, [0.2, 0.4, 0.8], [32, 64, 128], "cuboid"), (3, 777, 101, [0.35, 0.1], [48, 5], "ucube"),
                                                 (2, 300, 7, [5.0, 1e-4, 0.3], [400, 8, 300], "ucube"), (5, 5120, 510, [0.2], [32], "cuboid")])
def test_ball_query_all_radii_in_one_scan(oracle, ops, B, N, S, radii, Ks, dist):
    """[r5] PointNetSetAbstractionMsg's radius loop (models/pointnet2_utils.py:255-258) as one scan of the cloud: every list equals the
    single-radius call (the one-query-per-wave kernel, MP_BQ_LEGACY=1) and the oracle, bit for bit; ragged tiles (S % 4 != 0), full and
    empty balls, K beyond the cloud included."""
    import os
    from maskplanner_amd import synthetic as syn
    rng = np.random.default_rng(N + S + sum(Ks))
    xyz = syn.point_cloud(rng, B, N, dist)
    fidx = oracle.fps(xyz, S, rng.integers(0, N, size=B))
    new_xyz = oracle.index_points(xyz, fidx)
    got = ops.ball_query_multi(radii, Ks, dev(xyz), dev(new_xyz))
    os.environ["MP_BQ_LEGACY"] = "1"
    try:
        legacy = [ops.ball_query(r, K, dev(xyz), dev(new_xyz)) for r, K in zip(radii, Ks)]
    finally:
        del os.environ["MP_BQ_LEGACY"]
    for r, K, g, l in zip(radii, Ks, got, legacy):
        assert torch.equal(g, l), (r, K)
        if B * S * N <= 4 * 512 * 10240:
            assert np.array_equal(g.cpu().numpy(), oracle.ball_query(r, K, xyz, new_xyz)), (r, K)


@pytest.mark.parametrize("B,N,S,radii,Ks", [(1, 13312, 64, [0.3], [64]), (1, 10240, 64, [0.5], [512]), (1, 5120, 64, [0.9], [1024]), (1, 13312, 40, [0.3], [128]),
                                            (1, 13312, 32, [0.1, 0.2, 0.3], [32, 64, 128])])
def test_ball_query_at_the_lds_limits(oracle, ops, B, N, S, radii, Ks):
    """[r6] Shapes inside the documented range (N <= 13312, K <= 1024) whose hit lists do not fit beside the cloud with four queries per wave:
    the launch steps down to fewer lists per workgroup (and, for several radii, to one scan per radius) instead of refusing."""
    from maskplanner_amd import synthetic as syn
    rng = np.random.default_rng(N + sum(Ks))
    xyz = syn.point_cloud(rng, B, N, "cuboid")
    new_xyz = np.ascontiguousarray(xyz[:, :S])
    got = ops.ball_query_multi(radii, Ks, dev(xyz), dev(new_xyz)) if len(radii) > 1 else [ops.ball_query(radii[0], Ks[0], dev(xyz), dev(new_xyz))]
    for r, K, g in zip(radii, Ks, got):
        assert np.array_equal(g.cpu().numpy(), oracle.ball_query(r, K, xyz, new_xyz)), (r, K)


def test_ball_query_no_hit_returns_N(oracle, ops):
    xyz = np.random.default_rng(0).uniform(-1, 1, size=(1, 200, 3)).astype(np.float32)
    q = np.full((1, 3, 3), 50.0, np.float32)
    got = ops.ball_query(0.2, 8, dev(xyz), dev(q)).cpu().numpy()
    assert np.array_equal(got, oracle.ball_query(0.2, 8, xyz, q)) and (got == 200).all()


# ------------------------------------------------------------------------------------------------ gather / group
@pytest.mark.parametrize("det", [False, True])
def test_index_points_and_backward(oracle, ops, det):
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(3, 50, 7)).astype(np.float32)
    idx = rng.integers(0, 50, size=(3, 11, 6))
    p = dev(pts).requires_grad_(True)
    out = ops.index_points(p, dev(idx))
    assert np.array_equal(out.detach().cpu().numpy(), oracle.index_points(pts, idx))
    g = rng.normal(size=out.shape).astype(np.float32)
    ops.DETERMINISTIC = det
    try:
        out.backward(dev(g))
    finally:
        ops.DETERMINISTIC = False
    want = oracle.index_points_bwd(g.reshape(3, -1, 7), idx.reshape(3, -1), 50)
    got = p.grad.cpu().numpy()
    if det:
        assert np.array_equal(got, want)  # same summation order as the serial oracle
    else:
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("D,xyz_last", [(0, False), (13, False), (128, False), (13, True), (128, True), (64, True), (256, True)])
def test_group_and_backward(oracle, ops, D, xyz_last):
    rng = np.random.default_rng(6 + D)
    B, N, S, K = 2, 200, 16, 8
    xyz = rng.normal(size=(B, N, 3)).astype(np.float32)
    feats = rng.normal(size=(B, N, D)).astype(np.float32) if D else None
    new_xyz = xyz[:, :S].copy()
    idx = rng.integers(0, N, size=(B, S, K))
    want = oracle.group(xyz, feats, new_xyz, idx)
    if xyz_last:
        want = np.concatenate([want[..., 3:], want[..., :3]], -1)
    f = dev(feats).requires_grad_(True) if D else None
    out = ops.group(dev(xyz), f, dev(new_xyz), dev(idx), xyz_last=xyz_last)
    assert np.array_equal(out.detach().cpu().numpy(), want)
    if D:
        g = rng.normal(size=out.shape).astype(np.float32)
        gf = g[..., :D] if xyz_last else g[..., 3:]
        want_g = oracle.index_points_bwd(np.ascontiguousarray(gf).reshape(B, -1, D), idx.reshape(B, -1), N)
        for det in (False, True):
            ops.DETERMINISTIC = det
            try:
                f.grad = None
                ops.group(dev(xyz), f, dev(new_xyz), dev(idx), xyz_last=xyz_last).backward(dev(g))
            finally:
                ops.DETERMINISTIC = False
            if det:
                assert np.array_equal(f.grad.cpu().numpy(), want_g)
            else:
                np.testing.assert_allclose(f.grad.cpu().numpy(), want_g, rtol=1e-5, atol=1e-6)


def test_group_coordinates_only_padded_rows(oracle, ops):
    """The first level groups coordinates only, rows padded to 4 floats (group_xyz_kernel): [dx, dy, dz, 0] per neighbour, bit-exact,
    including out-of-range indices (clamped like the generic kernel) and a ragged last block of rows."""
    rng = np.random.default_rng(40)
    B, N, S, K = 3, 333, 37, 19
    xyz = rng.normal(size=(B, N, 3)).astype(np.float32)
    new_xyz = xyz[:, :S].copy()
    idx = rng.integers(0, N, size=(B, S, K))
    want = oracle.group(xyz, None, new_xyz, idx)
    out = ops.group(dev(xyz), None, dev(new_xyz), dev(idx), xyz_last=True, pad_to=4).cpu().numpy()
    assert out.shape == (B, S, K, 4)
    assert np.array_equal(out[..., :3], want) and not out[..., 3].any()


@pytest.mark.parametrize("D,S,K,every", [(64, 64, 32, 3), (128, 64, 32, 3), (256, 64, 32, 3), (128, 128, 64, 3), (128, 128, 64, 1),
                                         (320, 64, 32, 3), (192, 128, 64, 1), (320, 128, 64, 1)])   # 320 = the multi-scale level's concatenated features: column slabs
def test_group_internal_layout_fast_paths(oracle, ops, D, S, K, every):
    """The set-abstraction modules group with features first and rows padded to D + 4 floats: float4 row gather forward,
    atomic-free gather-reduce backward (csrc/group.hip).  One source point of cloud 1 is gathered by every `every`-th row:
    ~680 rows (fits its list), ~2700 rows (list + overflow list) or all 8192 rows (the workgroup's atomic fallback)."""
    rng = np.random.default_rng(60 + D + S)
    B, N = 3, 300
    xyz = rng.normal(size=(B, N, 3)).astype(np.float32)
    feats = rng.normal(size=(B, N, D)).astype(np.float32)
    new_xyz = xyz[:, :S].copy()
    idx = rng.integers(0, N, size=(B, S, K))
    idx[1].reshape(-1)[::every] = 7
    want = oracle.group(xyz, feats, new_xyz, idx)
    want = np.concatenate([want[..., 3:], want[..., :3], np.zeros(want.shape[:-1] + (1,), np.float32)], -1)
    f = dev(feats).requires_grad_(True)
    out = ops.group(dev(xyz), f, dev(new_xyz), dev(idx), xyz_last=True, pad_to=4)
    assert out.shape[-1] == D + 4 and np.array_equal(out.detach().cpu().numpy(), want)
    g = rng.normal(size=out.shape).astype(np.float32)
    want_g = oracle.index_points_bwd(np.ascontiguousarray(g[..., :D]).reshape(B, -1, D), idx.reshape(B, -1), N)
    out.backward(dev(g))
    got = f.grad.cpu().numpy()
    scale = np.abs(want_g).max()
    assert np.abs(got - want_g).max() <= 5e-5 * scale          # summation order differs (thousands of terms for the hot point)
    ops.DETERMINISTIC = True
    try:
        f.grad = None
        ops.group(dev(xyz), f, dev(new_xyz), dev(idx), xyz_last=True, pad_to=4).backward(dev(g))
        assert np.array_equal(f.grad.cpu().numpy(), want_g)
    finally:
        ops.DETERMINISTIC = False


# ------------------------------------------------------------------------------------------------ kNN
@pytest.mark.parametrize("B,P1,P2,D,K", [(3, 999, 985, 24, 1), (2, 3996, 2959, 6, 1), (2, 2959, 3996, 6, 1),
                                         (2, 700, 650, 3, 1), (2, 130, 70, 24, 2), (2, 64, 300, 5, 1), (1, 10, 3, 7, 4),
                                         (2, 257, 513, 12, 8), (2, 50, 1, 3, 1)])
def test_knn_vs_oracle(oracle, ops, B, P1, P2, D, K):
    rng = np.random.default_rng(P1 + P2 + D)
    p1 = rng.uniform(-1, 1, size=(B, P1, D)).astype(np.float32)
    p2 = rng.uniform(-1, 1, size=(B, P2, D)).astype(np.float32)
    l1 = rng.integers(max(1, P1 // 2), P1 + 1, size=B)
    l2 = rng.integers(max(1, P2 // 2), P2 + 1, size=B)
    l1[0], l2[0] = P1, P2
    wd, wi = oracle.knn_points(p1, p2, l1, l2, K)
    gd, gi = ops.knn(dev(p1), dev(p2), dev(l1), dev(l2), K)
    assert np.array_equal(gi.cpu().numpy(), wi)
    assert np.array_equal(gd.cpu().numpy().view(np.int32), wd.view(np.int32))
    # no lengths = full
    wd, wi = oracle.knn_points(p1, p2, None, None, K)
    gd, gi = ops.knn(dev(p1), dev(p2), None, None, K)
    assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gd.cpu().numpy(), wd)


def test_knn_ties_first_index(oracle, ops):
    lat = np.stack(np.meshgrid(np.arange(6), np.arange(6), np.arange(6), indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)
    q = lat[:, ::5] + 0.5  # equidistant from 8 lattice points
    wd, wi = oracle.knn_points(q, lat, None, None, 2)
    gd, gi = ops.knn(dev(q), dev(lat), None, None, 2)
    assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gd.cpu().numpy(), wd)


@pytest.mark.parametrize("det", [False, True])
@pytest.mark.parametrize("B,P1,P2,D,K", [(2, 300, 200, 24, 1), (2, 500, 900, 6, 1), (2, 40, 30, 3, 2)])
def test_knn_backward(oracle, ops, det, B, P1, P2, D, K):
    rng = np.random.default_rng(17 + D)
    p1 = rng.uniform(-1, 1, size=(B, P1, D)).astype(np.float32)
    p2 = rng.uniform(-1, 1, size=(B, P2, D)).astype(np.float32)
    l1 = np.array([P1, P1 - 7])
    l2 = np.array([P2 - 3, P2])
    g = rng.normal(size=(B, P1, K)).astype(np.float32)
    _, wi = oracle.knn_points(p1, p2, l1, l2, K)
    w1, w2 = oracle.knn_points_bwd(p1, p2, l1, l2, wi, g)
    a, b = dev(p1).requires_grad_(True), dev(p2).requires_grad_(True)
    ops.DETERMINISTIC = det
    try:
        d, _ = ops.knn(a, b, dev(l1), dev(l2), K)
        d.backward(dev(g))
    finally:
        ops.DETERMINISTIC = False
    np.testing.assert_allclose(a.grad.cpu().numpy(), w1, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(b.grad.cpu().numpy(), w2, rtol=1e-5, atol=2e-6)


def test_padded_lengths(oracle, ops):
    rng = np.random.default_rng(3)
    y = rng.normal(size=(5, 40, 6)).astype(np.float32)
    y[0, 17:] = -100.0
    y[1, 0:] = -100.0
    y[3, 39:] = -100.0
    y[4, 5, 0] = -100.0  # only the leading coordinate is tested by the reference
    assert np.array_equal(ops.padded_lengths(dev(y)).cpu().numpy(), oracle.padded_lengths(y))
    y2 = rng.normal(size=(2, 9, 3)).astype(np.float32)
    assert np.array_equal(ops.padded_lengths(dev(y2)).cpu().numpy(), [9, 9])


# ------------------------------------------------------------------------------------------------ mask matching
@pytest.mark.parametrize("B,M,S,n_ids", [(4, 6, 999, 6), (3, 22, 449, 15), (3, 41, 1266, 41), (2, 33, 1333, 20),
                                         (2, 6, 99, 9), (2, 64, 500, 64), (2, 5, 64, 1)])
@pytest.mark.parametrize("id_map", [(3.0, 1.0), (0.37, -0.63), (3.0, 5000.0)], ids=["whole", "fractional", "large"])
def test_mask_match_vs_scipy(oracle, ops, B, M, S, n_ids, id_map):
    """id_map: whole-number ids below 2048 take the kernel's bitmap route to the unique ids, anything else the general one."""
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(M * S)
    pred = (rng.normal(size=(B, M, S)) * 2).astype(np.float32)
    ids = rng.integers(0, n_ids, size=(B, S)).astype(np.float32) * np.float32(id_map[0]) + np.float32(id_map[1])
    match, uniq, nt, status, cost = ops.mask_match(dev(pred), dev(ids), return_cost=True)
    match, uniq, nt, status, cost = (t.cpu().numpy() for t in (match, uniq, nt, status, cost))
    assert (status == 0).all()
    for b in range(B):
        masks, u = oracle.stroke_ids_to_masks(ids[b])
        assert nt[b] == len(u) and np.array_equal(uniq[b, :len(u)], u)
        want_cost = oracle.mask_bce_cost(pred[b], masks)
        np.testing.assert_allclose(cost[b, :, :len(u)], want_cost, rtol=1e-6, atol=1e-4)
        # the assignment must be scipy's on the kernel's own fp32 cost (bit-exact index work) ...
        r, c = linear_sum_assignment(cost[b, :, :len(u)].astype(np.float64))
        want = np.full(M, -1)
        want[r] = c
        assert np.array_equal(match[b], want), b
        # ... and optimal for the oracle's cost as well
        r2, c2 = linear_sum_assignment(want_cost.astype(np.float64))
        tot = want_cost[r, c].sum()
        assert abs(tot - want_cost[r2, c2].sum()) <= 1e-5 * abs(tot)


def test_mask_match_smooth_targets_mse_cost(ops):
    """target_value given: cost[m,k] = sum_s (pred - value*[id == uid_k])^2 (loss_handler.py:810-811, 959-964)."""
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(77)
    B, M, S = 3, 22, 449
    pred = rng.normal(size=(B, M, S)).astype(np.float32)
    ids = rng.integers(0, 15, size=(B, S)).astype(np.float32)
    val = rng.uniform(0, 1, size=(B, S)).astype(np.float32)
    match, uniq, nt, status, cost = ops.mask_match(dev(pred), dev(ids), return_cost=True, target_value=dev(val))
    match, uniq, nt, cost = match.cpu().numpy(), uniq.cpu().numpy(), nt.cpu().numpy(), cost.cpu().numpy()
    assert (status.cpu().numpy() == 0).all()
    for b in range(B):
        u = np.unique(ids[b])
        assert nt[b] == len(u) and np.array_equal(uniq[b, :len(u)], u)
        tm = (ids[b][None, :] == u[:, None]) * val[b][None, :].astype(np.float64)                    # [Kb,S]
        want = ((pred[b][:, None, :].astype(np.float64) - tm[None]) ** 2).sum(-1)                     # [M,Kb]
        np.testing.assert_allclose(cost[b, :, :len(u)], want, rtol=1e-5)
        r, c = linear_sum_assignment(cost[b, :, :len(u)])
        got = match[b]
        assert np.array_equal(np.nonzero(got >= 0)[0], r) and np.array_equal(got[r], c)


def test_mask_match_failures_are_reported_and_poison_the_loss(ops):
    """What the reference asserts on (loss_handler.py:852: a predicted segment associated with the fake stroke id -1) or
    scipy raises for (non-finite costs), and more distinct strokes than the kernel holds, are reported per sample in `status`;
    the fused loss turns NaN instead of silently dropping the sample, and LossHandler.compute(return_list=True) raises."""
    from maskplanner_amd import loss_handler as LH
    rng = np.random.default_rng(3)
    B, M, S = 4, 6, 200
    pred = rng.normal(size=(B, M, S)).astype(np.float32)
    ids = rng.integers(0, 5, size=(B, S)).astype(np.float32)
    ids[1, 17] = -1.0                                   # padding id among the targets
    pred[2, 3, 11] = np.nan                             # invalid numeric entry in the cost
    big = rng.integers(0, 70, size=(S,)).astype(np.float32)
    big[:70] = np.arange(70)                            # 70 distinct strokes > MP_MASK_CAP
    ids[3] = big
    match, uniq, nt, status = ops.mask_match(dev(pred), dev(ids))
    st = status.cpu().numpy()
    assert st[0] == 0 and st[1] == ops.MATCH_PADDING_ID and st[2] == ops.MATCH_INFEASIBLE and st[3] == ops.MATCH_TOO_MANY_IDS, st
    m = match.cpu().numpy()
    assert (m[0] >= 0).sum() == 5 and (m[2] == -1).all() and (m[3] == -1).all()
    scores = dev(rng.normal(size=(B, M)).astype(np.float32))
    loss = ops.mask_loss(dev(pred), scores, dev(ids), match, uniq, 1.0, 100.0, 1.0, status=status)
    assert torch.isnan(loss)
    ok = ops.mask_loss(dev(pred[:1]), scores[:1], dev(ids[:1]), match[:1], uniq[:1], 1.0, 100.0, 1.0, status=status[:1])
    assert torch.isfinite(ok)
    # through the loss handler: the value is poisoned without a host sync, the synchronising call raises
    idx = torch.arange(S).repeat(B, 1).cuda()
    bad = LH.stroke_masks_loss(idx, dev(pred), scores, dev(ids), 1.0, 100.0, 1.0)
    assert torch.isnan(bad)
    with pytest.raises(AssertionError, match="stroke-mask matching failed"):
        LH.check_mask_matching()
    good = LH.stroke_masks_loss(idx[:1], dev(pred[:1]), scores[:1], dev(ids[:1]), 1.0, 100.0, 1.0)
    assert torch.isfinite(good)
    LH.check_mask_matching()


def test_mask_match_ties_follow_scipy(ops):
    """All-equal logits => a constant cost matrix: scipy's tie-breaking yields the identity."""
    from scipy.optimize import linear_sum_assignment
    B, M, S = 2, 7, 70
    pred = torch.zeros(B, M, S).cuda()
    ids = torch.arange(S).remainder(7).float()[None].repeat(B, 1).cuda()
    match, _, nt, status, cost = ops.mask_match(pred, ids, return_cost=True)
    c = cost[0, :, :7].cpu().numpy().astype(np.float64)
    r, cc = linear_sum_assignment(c)
    assert np.array_equal(match[0].cpu().numpy()[r], cc)
    # integer-valued costs with many ties
    rng = np.random.default_rng(9)
    pred = torch.from_numpy(rng.integers(-1, 2, size=(3, 9, 40)).astype(np.float32)).cuda()
    ids = torch.from_numpy(rng.integers(0, 5, size=(3, 40)).astype(np.float32)).cuda()
    match, _, nt, status, cost = ops.mask_match(pred, ids, return_cost=True)
    for b in range(3):
        k = int(nt[b])
        r, cc = linear_sum_assignment(cost[b, :, :k].cpu().numpy().astype(np.float64))
        want = np.full(9, -1)
        want[r] = cc
        assert np.array_equal(match[b].cpu().numpy(), want)


# ------------------------------------------------------------------------------------------------ 3-NN interpolation
def test_three_nn_golden_bit_exact(golden, ops):
    g = golden("g10_fp")
    for t in "ab":
        idx, w, d = ops.three_nn(dev(g[t + "_xyz1"]), dev(g[t + "_xyz2"]), return_dist=True)
        assert np.array_equal(idx.cpu().numpy(), g[t + "_nn_idx"]), t
        assert np.array_equal(d.cpu().numpy().view(np.int32), g[t + "_nn_dist"].view(np.int32)), t
        assert np.array_equal(w.cpu().numpy().view(np.int32), g[t + "_nn_weight"].view(np.int32)), t
        p2 = dev(g[t + "_points2"].transpose(0, 2, 1))
        out = ops.three_interpolate(p2, idx, w)
        assert np.array_equal(out.cpu().numpy().view(np.int32), g[t + "_interp"].view(np.int32)), t


@pytest.mark.parametrize("B,N,S,D", [(32, 5120, 1024, 64), (4, 1024, 256, 128), (3, 777, 3, 5), (2, 3000, 2500, 1030), (2, 1, 5, 4)])
def test_three_nn_vs_oracle(oracle, ops, B, N, S, D):
    from maskplanner_amd import synthetic as syn
    rng = np.random.default_rng(N + S)
    xyz1 = syn.point_cloud(rng, B, N, "cuboid")
    # sources = a subset of the queries (coincident points) when possible, else independent points
    xyz2 = xyz1[:, rng.permutation(N)[:S]].copy() if S <= N else rng.uniform(-1, 1, (B, S, 3)).astype(np.float32)
    d0, i0, w0 = oracle.three_nn(xyz1, xyz2)
    idx, w, d = ops.three_nn(dev(xyz1), dev(xyz2), return_dist=True)
    assert np.array_equal(d.cpu().numpy().view(np.int32), d0.view(np.int32))   # distances first: indices may tie
    ties = (d0[..., 0] == d0[..., 1]) | (d0[..., 1] == d0[..., 2])
    assert np.array_equal(idx.cpu().numpy()[~ties], i0[~ties])
    assert np.array_equal(idx.cpu().numpy(), i0)                               # same first-index rule on ties
    assert np.array_equal(w.cpu().numpy().view(np.int32), w0.view(np.int32))
    p2 = rng.normal(size=(B, S, D)).astype(np.float32)
    out = ops.three_interpolate(dev(p2), idx, w)
    assert np.array_equal(out.cpu().numpy().view(np.int32), oracle.three_interpolate(p2, i0, w0).view(np.int32))


@pytest.mark.parametrize("det", [False, True])
def test_three_interpolate_backward(oracle, ops, det):
    import maskplanner_amd.ops as O
    rng = np.random.default_rng(11)
    B, N, S, D = 3, 900, 70, 37
    xyz1 = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    xyz2 = rng.uniform(-1, 1, (B, S, 3)).astype(np.float32)
    _, i0, w0 = oracle.three_nn(xyz1, xyz2)
    p2 = dev(rng.normal(size=(B, S, D)).astype(np.float32)).requires_grad_(True)
    go = rng.normal(size=(B, N, D)).astype(np.float32)
    old = O.DETERMINISTIC
    O.DETERMINISTIC = det
    try:
        out = ops.three_interpolate(p2, dev(i0), dev(w0))
        (g,) = torch.autograd.grad(out, [p2], dev(go))
        g = g.cpu().numpy()
        want = oracle.three_interpolate_bwd(go, i0, w0, S)
        if det:
            assert np.array_equal(g.view(np.int32), want.view(np.int32))       # same summation order as the oracle
            (g2,) = torch.autograd.grad(ops.three_interpolate(p2, dev(i0), dev(w0)), [p2], dev(go))
            assert np.array_equal(g2.cpu().numpy(), g)
        else:
            np.testing.assert_allclose(g, want, rtol=1e-4, atol=1e-4)
    finally:
        O.DETERMINISTIC = old


def test_three_nn_rejects_bad_input(ops):
    with pytest.raises(ValueError):
        ops.three_nn(torch.zeros(1, 4, 3, device="cuda"), torch.zeros(1, 2, 3, device="cuda"))
    with pytest.raises((ValueError, RuntimeError)):
        ops.three_nn(torch.zeros(1, 4, 3), torch.zeros(1, 5, 3))   # CPU tensors are refused


# ------------------------------------------------------------------------------------------------ large LAP (segment matching)
def _scipy_pairs(c):
    from scipy.optimize import linear_sum_assignment
    return linear_sum_assignment(c)


@pytest.mark.parametrize("shapes", [[(999, 900), (999, 985), (999, 999)], [(40, 70), (70, 40), (1, 5), (5, 1), (64, 64)],
                                    [(1333, 1290), (449, 500)], [(300, 2048), (2048, 300)]])
def test_lsap_matches_scipy(ops, shapes):
    """hungarianMatcher.py:58-61: same assignment as scipy.optimize.linear_sum_assignment (continuous random costs: the
    optimum is unique), rows ascending, rectangular both ways, ragged batch."""
    rng = np.random.default_rng(sum(a * b for a, b in shapes))
    costs = [rng.uniform(0, 3, size=s).astype(np.float32) for s in shapes]
    pairs, status = ops.lsap([dev(c) for c in costs])
    assert (status.cpu().numpy() == 0).all()
    for c, (i, j) in zip(costs, pairs):
        r, k = _scipy_pairs(c)
        assert i.dtype == torch.int64 and j.dtype == torch.int64
        assert np.array_equal(i.cpu().numpy(), r) and np.array_equal(j.cpu().numpy(), k), c.shape


def test_lsap_ties_follow_scipy(ops):
    """Integer-valued costs are full of ties: the scan-order rules of scipy's rectangular_lsap decide, and are reproduced."""
    rng = np.random.default_rng(3)
    costs = [rng.integers(0, 4, size=s).astype(np.float32) for s in [(30, 30), (50, 80), (80, 50), (200, 210), (7, 7)]]
    costs.append(np.zeros((20, 25), np.float32))
    pairs, status = ops.lsap([dev(c) for c in costs])
    assert (status.cpu().numpy() == 0).all()
    for c, (i, j) in zip(costs, pairs):
        r, k = _scipy_pairs(c)
        assert np.array_equal(i.cpu().numpy(), r) and np.array_equal(j.cpu().numpy(), k), c.shape


@pytest.mark.parametrize("shapes,hi", [([(400, 1000), (999, 999), (130, 129)], 3), ([(300, 2048), (1500, 700)], 2), ([(512, 512), (256, 600)], 6)])
def test_lsap_ties_at_the_sizes_of_the_four_wave_kernel(ops, shapes, hi):
    """[r4] More than 128 columns: four waves per sample, column state in registers, one candidate per wave merged behind a barrier
    (lsap4_kernel).  Small integer costs at 512 / 1024 / 2048 columns: almost every Dijkstra step is decided by scipy's scan-order
    rules, across the waves too; a share of +inf entries (forbidden pairs that leave the problem feasible) rides along."""
    rng = np.random.default_rng(hi + len(shapes))
    costs = [rng.integers(0, hi, size=s).astype(np.float32) for s in shapes]
    holes = costs[0].copy()
    holes[rng.uniform(size=holes.shape) < 0.3] = np.inf
    costs.append(holes)
    dup = rng.uniform(0, 1, size=(300, 1, 3)).astype(np.float32).repeat(3, 1).reshape(900, 3)      # every target three times: tied columns
    costs.append(np.sqrt(((rng.uniform(0, 1, size=(700, 1, 3)).astype(np.float32) - dup[None]) ** 2).sum(-1)).astype(np.float32))
    pairs, status = ops.lsap([dev(c) for c in costs])
    assert (status.cpu().numpy() == 0).all()
    for c, (i, j) in zip(costs, pairs):
        r, k = _scipy_pairs(c)
        assert np.array_equal(i.cpu().numpy(), r) and np.array_equal(j.cpu().numpy(), k), c.shape


def test_lsap_infeasible_and_empty(ops):
    c = np.full((3, 3), np.inf, np.float32)
    pairs, status = ops.lsap([dev(c), dev(np.eye(3, dtype=np.float32))])
    st = status.cpu().numpy()
    assert st[0] != 0 and st[1] == 0
    assert (pairs[0][1].cpu().numpy() == -1).all()
    assert np.array_equal(pairs[1][1].cpu().numpy(), np.array([1, 0, 2])) or np.array_equal(pairs[1][1].cpu().numpy(), _scipy_pairs(np.eye(3))[1])
    assert ops.lsap([]) == ([], None)


# ------------------------------------------------------------------------------------------------ chamfer reductions
@pytest.mark.parametrize("point,batch", [("mean", "mean"), ("sum", "mean"), ("mean", "sum"), ("sum", None), ("mean", None)])
def test_chamfer_reduce_matches_torch_algebra(ops, point, batch):
    """pytorch3d_chamfer.py:295-326 as one kernel forward / one backward, against the same algebra in torch (fp64)."""
    rng = np.random.default_rng(17)
    N, P = 32, 999
    lengths = rng.integers(500, P + 1, size=N)
    cham = rng.uniform(0, 2, size=(N, P)).astype(np.float32)
    cham[np.arange(P)[None] >= lengths[:, None]] = 0.0            # what the kNN kernel leaves beyond a cloud's length
    c = dev(cham).requires_grad_(True)
    scale = 2.5 if point == "mean" else 1.0                        # constant factor folded into the kernel
    out = ops.chamfer_reduce(c, dev(lengths), point, batch, scale)
    c64 = torch.from_numpy(cham).double().requires_grad_(True)
    ref = c64.sum(1) * scale
    if point == "mean":
        ref = ref / torch.from_numpy(lengths)
    if batch is not None:
        ref = ref.sum()
        if batch == "mean":
            ref = ref / N
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-6)
    g = rng.normal(size=tuple(ref.shape)).astype(np.float32)
    out.backward(torch.tensor(g).cuda())
    ref.backward(torch.tensor(g).double())
    want = c64.grad.numpy().copy()
    want[np.arange(P)[None] >= lengths[:, None]] = 0.0            # no gradient into the padded rows
    np.testing.assert_allclose(c.grad.cpu().numpy(), want, rtol=2e-6, atol=1e-12)
    out2 = ops.chamfer_reduce(dev(cham), dev(lengths), point, batch, scale)
    assert torch.equal(out2, out.detach())                          # deterministic


# ------------------------------------------------------------------------------------------------ fused tails
def test_pose_output_matches_torch_algebra(ops):
    """models/pointnet2_cls_ssg.py:332-339 as one kernel, against the reference's op chain (values and both gradients);
    includes an all-zero raw normal (tanh(0) = 0: the F.normalize eps branch)."""
    import torch.nn.functional as F
    rng = np.random.default_rng(23)
    B, S, lam, w = 4, 99, 4, 0.25
    pos = rng.normal(size=(B, S * lam * 3)).astype(np.float32)
    raw = rng.normal(size=(B, S * lam * 3)).astype(np.float32) * 2
    raw[0, :3] = 0.0
    p1, r1 = dev(pos).requires_grad_(True), dev(raw).requires_grad_(True)
    out = ops.pose_output(p1, r1, w).view(B, S, -1)
    p2, r2 = dev(pos).requires_grad_(True), dev(raw).requires_grad_(True)
    normals = F.normalize(torch.tanh(r2).view(B, -1, 3), dim=-1) * w
    ref = torch.cat((p2.view(B, -1, 3), normals), dim=-1).view(B, S, -1)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-6, atol=1e-7)
    g = dev(rng.normal(size=tuple(ref.shape)).astype(np.float32))
    out.backward(g)
    ref.backward(g)
    np.testing.assert_allclose(p1.grad.cpu().numpy(), p2.grad.cpu().numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(r1.grad.cpu().numpy(), r2.grad.cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,M,S,n_ids,nsw", [(4, 6, 999, 6, 1.0), (3, 22, 449, 15, 0.5), (2, 41, 1266, 30, 0.1)])
def test_mask_loss_matches_torch_algebra(ops, B, M, S, n_ids, nsw):
    """loss_handler.py:877-934 (binary targets) fused, against the same algebra in torch ops."""
    import torch.nn.functional as F
    rng = np.random.default_rng(B * M + S)
    pred = rng.normal(size=(B, M, S)).astype(np.float32)
    scores = rng.normal(size=(B, M)).astype(np.float32)
    ids = rng.integers(0, n_ids, size=(B, S)).astype(np.float32)
    match, uniq, nt, status = ops.mask_match(dev(pred), dev(ids))
    w_masks, w_conf = 1.0, 100.0
    a, sa = dev(pred).requires_grad_(True), dev(scores).requires_grad_(True)
    out = ops.mask_loss(a, sa, dev(ids), match, uniq, w_masks, w_conf, nsw)
    b, sb = dev(pred).requires_grad_(True), dev(scores).requires_grad_(True)
    matched = match >= 0
    uid = uniq.gather(1, match.clamp(min=0))
    tm = (dev(ids)[:, None, :] == uid[:, :, None]).float()
    per_mask = F.binary_cross_entropy_with_logits(b, tm, reduction="none").sum(-1)
    mask_loss = (per_mask * matched).sum() / matched.sum()
    weights = torch.where(matched, torch.ones_like(sb), torch.full_like(sb, nsw))
    conf = F.binary_cross_entropy_with_logits(sb, matched.float(), weight=weights, reduction="none").mean()
    ref = w_masks * mask_loss + w_conf * conf
    np.testing.assert_allclose(float(out), float(ref), rtol=2e-6)
    out.backward()
    ref.backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(sa.grad.cpu().numpy(), sb.grad.cpu().numpy(), rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("B,C", [(32, 1024), (4, 64), (2, 1000), (64, 1024), (50, 200), (100, 64)])
def test_bn_relu_rows_matches_batchnorm1d(ops, train, B, C):
    """F.relu(nn.BatchNorm1d(x)) fused for skinny batches: outputs, gradients and running statistics against torch's own."""
    import torch.nn.functional as F
    torch.manual_seed(B * C)
    bn_a, bn_b = torch.nn.BatchNorm1d(C).cuda(), torch.nn.BatchNorm1d(C).cuda()
    with torch.no_grad():
        for bn in (bn_a, bn_b):
            bn.weight.copy_(torch.linspace(0.5, 1.5, C))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
            bn.running_mean.copy_(torch.linspace(-0.2, 0.2, C))
            bn.running_var.copy_(torch.linspace(0.5, 1.5, C))
    bn_a.train(train)
    bn_b.train(train)
    x = torch.randn(B, C, device="cuda") * 2 + 0.5
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = ops.bn_relu_rows(xa, bn_a)
    yb = F.relu(bn_b(xb))
    np.testing.assert_allclose(ya.detach().cpu().numpy(), yb.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    g = torch.randn(B, C, device="cuda")
    ya.backward(g)
    yb.backward(g)
    np.testing.assert_allclose(xa.grad.cpu().numpy(), xb.grad.cpu().numpy(), rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(bn_a.weight.grad.cpu().numpy(), bn_b.weight.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(bn_a.bias.grad.cpu().numpy(), bn_b.bias.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(bn_a.running_mean.cpu().numpy(), bn_b.running_mean.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bn_a.running_var.cpu().numpy(), bn_b.running_var.cpu().numpy(), rtol=1e-5, atol=1e-6)



def test_match_segments_cost_layout_and_assignment(oracle):
    """mp_cdist_batch_f32 + mp_lsap_f32 (ops.match_segments): per-sample Euclidean cost blocks in one launch, transposed where a
    sample has more predictions than targets, against the oracle's cdist and scipy on it -- more targets than predictions, fewer,
    equal, and an empty target list."""
    from scipy.optimize import linear_sum_assignment
    from maskplanner_amd import ops
    rng = np.random.default_rng(12)
    B, S, D = 4, 40, 24
    x = rng.normal(size=(B, S, D)).astype(np.float32)
    sizes = [55, 40, 17, 33]
    ys = [rng.normal(size=(t, D)).astype(np.float32) for t in sizes]
    pairs, status = ops.match_segments(dev(x), [dev(y) for y in ys])
    assert (status.cpu().numpy() == 0).all()
    for b, (i, j) in enumerate(pairs):
        c = oracle.cdist(x[b], ys[b])
        ri, cj = linear_sum_assignment(c.astype(np.float64))
        assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(j.cpu().numpy(), cj), b


@pytest.mark.gpu
def test_chamfer_term_equals_knn_plus_reduction():
    """ops.chamfer_term = one reduced one-directional chamfer term (knn K=1 + reduction forward, a single knn_bwd launch with the
    row gradient formed in the kernel).  Same value, distances, indices and -- bit for bit with the fixed-order scatter -- the same
    gradients as ops.knn followed by ops.chamfer_reduce, for ragged lengths on both sides, every reduction mode and a chained `add`."""
    from maskplanner_amd import ops
    torch.manual_seed(3)
    B, P1, P2, D = 5, 257, 190, 6
    l1 = torch.tensor([257, 100, 1, 200, 64], device="cuda")
    l2 = torch.tensor([190, 3, 190, 77, 128], device="cuda")
    det, ops.DETERMINISTIC = ops.DETERMINISTIC, True
    try:
        for pr, br in (("mean", "mean"), ("sum", "mean"), ("mean", "sum"), ("mean", None)):
            x = torch.randn(B, P1, D, device="cuda", requires_grad=True)
            y = torch.randn(B, P2, D, device="cuda", requires_grad=True)
            add = torch.tensor(0.37, device="cuda", requires_grad=True) if br is not None else None
            d, i = ops.knn(x, y, l1, l2, 1)
            want = ops.chamfer_reduce(d.view(B, P1), l1, pr, br, 2.5, add=add)
            w = torch.randn_like(want)
            (want * w).sum().backward()
            gx, gy, ga = x.grad.clone(), y.grad.clone(), (None if add is None else add.grad.clone())
            x.grad = y.grad = None
            if add is not None:
                add.grad = None
            got, dd, ii = ops.chamfer_term(x, y, l1, l2, pr, br, 2.5, add=add)
            (got * w).sum().backward()
            assert torch.equal(got, want) and torch.equal(dd, d.view(B, P1)) and torch.equal(ii, i.view(B, P1)), (pr, br)
            assert torch.equal(x.grad, gx) and torch.equal(y.grad, gy), (pr, br, float((x.grad - gx).abs().max()), float((y.grad - gy).abs().max()))
            assert add is None or torch.equal(add.grad, ga)
    finally:
        ops.DETERMINISTIC = det


@pytest.mark.gpu
def test_fused_dropout_of_the_head_blocks():
    """ops.bn_relu_rows(dropout=(p, rng, layer)): nn.Dropout(p) inside the BatchNorm + ReLU launch.  Kept elements equal the
    undropped output / (1 - p), dropped ones are exact zeros, the keep rate is 1 - p, masks differ between steps and layers and
    repeat for the same (seed, step, layer); the backward is the undropped backward of the masked, rescaled gradient."""
    from maskplanner_amd import ops
    torch.manual_seed(0)
    B, C, p = 32, 1024, 0.3
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    x = torch.randn(B, C, device="cuda", requires_grad=True)
    rng = torch.tensor([1234567, 0], dtype=torch.int64, device="cuda")
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    base = ops.bn_relu_rows(x, bn)
    bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    y0 = ops.bn_relu_rows(x, bn, dropout=(p, rng, 0))
    kept = y0 != 0
    pos = base > 0
    assert not bool((kept & ~pos).any())
    rate = float((kept & pos).sum()) / float(pos.sum())
    assert abs(rate - (1 - p)) < 0.01, rate
    assert torch.allclose(y0[kept], base[kept] / (1 - p), rtol=1e-6, atol=0)
    # same (seed, step, layer): the same mask; another layer, another step: other masks
    assert torch.equal(ops.bn_relu_rows(x, bn, dropout=(p, rng, 0)) != 0, kept)
    other_layer = ops.bn_relu_rows(x, bn, dropout=(p, rng, 1)) != 0
    rng[1] += 1
    other_step = ops.bn_relu_rows(x, bn, dropout=(p, rng, 0)) != 0
    for m in (other_layer, other_step):
        agree = float(((m == kept) & pos).sum()) / float(pos.sum())
        assert abs(agree - ((1 - p) ** 2 + p ** 2)) < 0.02, agree     # independent Bernoulli masks
    rng[1] -= 1
    # backward: d/dx of sum(w * dropout(relu(bn(x)))) == the undropped block's backward fed with w * mask / (1 - p)
    w = torch.randn(B, C, device="cuda")
    y0 = ops.bn_relu_rows(x, bn, dropout=(p, rng, 0))
    (y0 * w).sum().backward()
    gx, gg, gb = x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()
    x.grad = None; bn.weight.grad = None; bn.bias.grad = None
    base = ops.bn_relu_rows(x, bn)
    (base * (w * kept / (1 - p))).sum().backward()
    assert torch.allclose(x.grad, gx, rtol=1e-5, atol=1e-6) and torch.allclose(bn.weight.grad, gg, rtol=1e-5, atol=1e-5)
    assert torch.allclose(bn.bias.grad, gb, rtol=1e-5, atol=1e-5)
    # p = 0 keeps everything; eval mode is untouched by the harness (the model only takes this path in train mode)
    assert torch.equal(ops.bn_relu_rows(x, bn, dropout=(0.0, rng, 0)), ops.bn_relu_rows(x, bn))


@pytest.mark.gpu
@pytest.mark.parametrize("D,P1,P2", [(6, 1500, 3996), (24, 999, 700), (3, 700, 2048), (12, 300, 517)])
def test_screened_nearest_neighbour_equals_the_direct_scan(D, P1, P2):
    """mp_knn1_f32(screened=1): the bf16 matrix cores screen the pairs through three-plane split dot products and only the rows inside
    the error window of the minimum are evaluated with the direct kernel's arithmetic.  The result must be the direct scan's, bit for bit:
    distances, indices, first index on exact ties -- on random clouds, duplicated references (ties inside and across 32-row blocks),
    queries that coincide with references (zero distances), collapsed references (every row inside the window: the full-scan path),
    references far from the origin (large norms, small distances) and ragged lengths on both sides."""
    from maskplanner_amd import _lib, ops
    lib = _lib.load()

    def run(p1, p2, l1, l2, screened):
        B = p1.shape[0]
        d = torch.empty(B, p1.shape[1], device="cuda")
        i = torch.empty(B, p1.shape[1], dtype=torch.int64, device="cuda")
        nb = lib.mp_knn1_workspace_bytes(B, p2.shape[1], D)
        assert nb > 0
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        ops._run("knn1", p1, lib.mp_knn1_f32, p1.data_ptr(), p2.data_ptr(), None if l1 is None else l1.data_ptr(),
                 None if l2 is None else l2.data_ptr(), B, p1.shape[1], p2.shape[1], D, d.data_ptr(), i.data_ptr(), int(screened),
                 ws.data_ptr(), ws.numel())
        return d, i

    torch.manual_seed(D)
    B = 6
    p1 = torch.rand(B, P1, D, device="cuda")
    p2 = torch.rand(B, P2, D, device="cuda")
    p2[0, 5] = p2[0, 3]                       # a tie inside a block
    p2[0, 200:264] = p2[0, 100:164]           # ties across blocks and tiles
    p1[1, 7] = p2[1, 11]                      # zero distance
    p2[2] = p2[2, :1] + 1e-7 * torch.randn(P2, D, device="cuda")        # collapsed references: everything inside the window
    p2[3] = p2[3] + 40.0                      # far from the origin: norms ~1e4, distances ~1
    p1[3] = p1[3] + 40.0
    p2[4] = p2[4, torch.randint(0, 8, (P2,), device="cuda")]            # eight distinct rows repeated all over: more ties than records
    l1 = torch.tensor([P1, P1 // 2, P1, 17, P1, 1], device="cuda")
    l2 = torch.tensor([P2, P2, P2 - 3, P2, P2, 33], device="cuda")
    for a, bb in ((l1, l2), (None, None)):
        d0, i0 = run(p1, p2, a, bb, 0)
        d1, i1 = run(p1, p2, a, bb, 1)
        assert torch.equal(i0, i1), (int((i0 != i1).sum()), (i0 != i1).nonzero()[:4].tolist())
        assert torch.equal(d0, d1), int((d0 != d1).sum())
    # the default entry point takes the screened path when it is handed the workspace: same result through ops.knn
    dd, ii = ops.knn(p1, p2, l1, l2, 1)
    d0, i0 = run(p1, p2, l1, l2, 0)
    assert torch.equal(dd[..., 0], d0) and torch.equal(ii[..., 0], i0)


@pytest.mark.parametrize("B,O,I", [(32, 11988, 1024), (32, 5994, 1024), (7, 4100, 256), (32, 4097, 128), (1, 16, 128), (64, 11988, 1024), (45, 4100, 256)])
def test_head_input_gradient_kernels_vs_fp64(ops, B, O, I):
    """grad_x = g W of the wide head Linears (autograd of nn.Linear, models/pointnet2_cls_ssg.py:311, 327, 336): the matrix-core form
    (csrc/linear_dx.hip: three bf16 planes per fp32 operand) and the ordered VALU form against an fp64 product -- both at fp32
    rounding level, ragged row counts (O % 16 != 0) and short batches included."""
    from maskplanner_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(O + B)
    g = torch.randn(B, O, generator=gen).cuda()
    W = (torch.randn(O, I, generator=gen) / 32).cuda()
    want = g.double() @ W.double()
    scale = float(want.abs().max())
    gx = torch.full((B, I), float("nan"), device="cuda")
    ops._run("linear_dx_mfma", g, lib.mp_linear_dx_mfma_f32, g.data_ptr(), W.data_ptr(), B, O, I, gx.data_ptr())
    err_m = float((gx.double() - want).abs().max()) / scale
    from maskplanner_amd import factor_heads as fh
    gy = fh._dx_skinny(g, W)       # (32 batch rows per pass over W)
    err_s = float((gy.double() - want).abs().max()) / scale
    assert err_m < 2e-6 and err_s < 2e-6, (err_m, err_s)
    # (the matrix-core form adds its K slices with atomics: its error moves with their order from run to run)
    assert err_m < 4.0 * err_s + 4e-7, (err_m, err_s)


def test_static_target_lengths_and_prepared_planes(ops):
    """ops.register_static_target: padded_lengths() returns the registered lengths without a launch, and a chamfer term whose reference
    set is the registered tensor searches against the prepared planes (mp_knn1_prepare_f32 / mp_knn1_prepared_f32) -- bit-identical to
    the unregistered path; refresh_static_target follows a rewrite of the buffer."""
    torch.manual_seed(5)
    B, P1, P2, D = 4, 300, 700, 24
    y = torch.randn(B, P2, D, device="cuda")
    y[1, 500:] = -100.0
    y[3, 256:] = -100.0
    x = torch.randn(B, P1, D, device="cuda", requires_grad=True)
    len1 = torch.full((B,), P1, dtype=torch.int64, device="cuda")
    ref_len = ops.padded_lengths(y)
    ref = ops.chamfer_term(x, y, len1, ref_len, "mean", "mean", 3.0)
    try:
        e = ops.register_static_target(y, planes=True)
        assert ops.padded_lengths(y) is e["lengths"] and torch.equal(e["lengths"], ref_len)
        got = ops.chamfer_term(x, y, len1, ops.padded_lengths(y), "mean", "mean", 3.0)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
        y[1, 400:] = -100.0
        y[0] += 0.25
        ops.refresh_static_target(y)
        want = ops.chamfer_term(x, y.clone(), len1, ops.padded_lengths(y.clone()), "mean", "mean", 3.0)
        got = ops.chamfer_term(x, y, len1, ops.padded_lengths(y), "mean", "mean", 3.0)
        assert torch.equal(got[0], want[0]) and torch.equal(got[2], want[2])
        g, = torch.autograd.grad(got[0], x)
        gw, = torch.autograd.grad(want[0], x)
        assert torch.equal(g, gw)
    finally:
        ops.forget_static_targets()


def test_grad_accum_shared_buffer_equals_separate_gradients(ops):
    """[r5, ADVICE r4] ops.GradAccum outside the full training step: three chamfer terms on one prediction (as x, as y, as x again) with an armed
    ZeroArena -- the gradient handed to autograd by the LAST term to run equals the sum of the three separate gradients; a term whose shared
    operand takes no gradient (a detached alias) is not counted; and two arenas on two streams do not interfere."""
    torch.manual_seed(3)
    B, P, Q, D = 3, 200, 150, 6
    tgt = torch.randn(B, Q, D, device="cuda")
    lp = torch.full((B,), P, dtype=torch.int64, device="cuda")
    lq = torch.full((B,), Q, dtype=torch.int64, device="cuda")
    base = torch.randn(B, P, D, device="cuda")

    def loss(pred, acc):
        a = ops.chamfer_term(pred, tgt, lp, lq, "mean", "mean", 1.0, grad_accum=acc)[0]
        b = ops.chamfer_term(tgt, pred, lq, lp, "mean", "mean", 2.0, grad_accum=acc)[0]
        c = ops.chamfer_term(pred, tgt, lp, lq, "sum", "mean", 0.5, grad_accum=acc)[0]
        d = ops.chamfer_term(pred.detach(), tgt, lp, lq, "mean", "mean", 3.0, grad_accum=acc)[0]      # same storage, no gradient: not counted
        return a + b + c + d

    ref = base.clone().requires_grad_(True)
    loss(ref, None).backward()
    arena = ops.ZeroArena(torch.device("cuda", 0))
    for _ in range(2):          # (the first pass teaches the arena its size)
        got = base.clone().requires_grad_(True)
        acc = ops.GradAccum(got)
        val = loss(got, acc)
        arena.arm()
        try:
            val.backward()
        finally:
            arena.disarm()
    assert acc.pending == 0 and acc.buf is not None
    assert float((got.grad - ref.grad).abs().max()) <= 1e-6 * float(ref.grad.abs().max())
    # one owner per stream, independent streams
    other = ops.ZeroArena(torch.device("cuda", 0))
    arena.arm()
    with pytest.raises(RuntimeError, match="another ZeroArena"):
        other.arm()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        other.arm()
        assert ops.ZeroArena.current(torch.device("cuda", 0)) is other
        other.disarm()
    assert ops.ZeroArena.current(torch.device("cuda", 0)) is arena
    arena.disarm()
    assert ops.ZeroArena.current(torch.device("cuda", 0)) is None
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_profiler_marks_time_a_kernel_inside_a_replayed_graph():
    """mp_profiler_mark / mp_profiler_read_marks (bench.py's roofline timing): a marked launch recorded into a hipGraph is timestamped by every
    replay; the mean over the last n replays is positive, close to the HIP-event figure of eager launches, and unmarked launches leave nothing."""
    import ctypes
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from maskplanner_amd import _lib, ops
    lib = _lib.load()
    B, N, S = 8, 4096, 256
    xyz = torch.rand(B, N, 3, device="cuda")
    start = torch.zeros(B, dtype=torch.long, device="cuda")
    ops.fps(xyz, S, start)                                  # first launch outside the recording (dynamic-LDS opt-in)
    torch.cuda.synchronize()

    def read(n):
        buf = ctypes.create_string_buffer(1 << 12)
        k = lib.mp_profiler_read_marks(buf, len(buf), n)
        assert k >= 0
        return {l.split("\t")[0]: (int(l.split("\t")[1]), float(l.split("\t")[2])) for l in buf.value.decode().strip().split("\n") if l}

    try:
        assert lib.mp_profiler_mark(b"no_such_kernel|fps_kernel") == 0
        g = torch.cuda.CUDAGraph()
        from maskplanner_amd.harness import recording
        with recording(g):
            idx = ops.fps(xyz, S, start)
            ops.ball_query(0.2, 16, xyz, xyz[:, :64].contiguous())     # not marked
        for _ in range(6):
            g.replay()
        got = read(4)
        assert list(got) and all(k.startswith("fps_kernel") for k in got), got
        (calls, ms), = got.values()
        assert calls == 4 and ms > 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            ops.fps(xyz, S, start)
        e1.record()
        torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) / 4
        assert 0.5 * eager < ms / calls < 1.5 * eager + 0.02, (ms / calls, eager)
        assert torch.equal(idx, ops.fps(xyz, S, start))
    finally:
        lib.mp_profiler_mark(None)
    assert read(4) == {}
