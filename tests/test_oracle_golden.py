"""The CPU oracle against the golden vectors produced by the imported reference (oracle/gen_golden.py).

This is what pins the oracle: bit-exact for index work (FPS, ball query, LAP), 1e-5 for fp32 values.
"""
import numpy as np
import pytest
import torch

from oracle import torch_ref as T

RTOL = 1e-5  # north_star tolerance for fp32 values


def _cases(g, suffix):
    return sorted({k[: -len(suffix)] for k in g.files if k.endswith(suffix)})


def test_fps_bit_exact(golden, oracle):
    g = golden("g1_fps")
    names = _cases(g, "_idx")
    assert len(names) >= 5
    for k in names:
        idx = g[k + "_idx"]
        out = oracle.fps(g[k + "_xyz"], idx.shape[1], g[k + "_start"])
        assert out.dtype == np.int64 and np.array_equal(out, idx), k


def test_square_distance_bit_exact(golden, oracle):
    g = golden("g2_sqd")
    for t in "ab":
        out = oracle.square_distance(g[t + "_src"], g[t + "_dst"])
        assert np.array_equal(out.view(np.int32), g[t + "_out"].view(np.int32))


def test_ball_query_bit_exact(golden, oracle):
    g = golden("g2_bq")
    names = _cases(g, "_idx")
    assert len(names) >= 7
    for k in names:
        out = oracle.ball_query(float(g[k + "_radius"]), int(g[k + "_K"]), g[k + "_xyz"], g[k + "_new_xyz"])
        assert np.array_equal(out, g[k + "_idx"].astype(np.int64)), k


def test_ball_query_threshold_cast(oracle):
    """radius**2 is squared in double, then cast to f32 (NOT f32(r)*f32(r)): pointnet2_utils.py:104."""
    assert np.float32(0.2 * 0.2) != np.float32(0.2) * np.float32(0.2)
    r2f = np.float32(0.2 * 0.2)
    # a point at squared distance exactly r2f is inside; one ulp above is outside
    q = np.zeros((1, 1, 3), np.float32)
    inside = np.array([np.sqrt(np.float64(r2f)), 0, 0])
    xyz = np.zeros((1, 4, 3), np.float32)
    xyz[0, 1, 0] = np.float32(inside[0])
    xyz[0, 2, 0] = 5.0
    xyz[0, 3, 0] = np.nextafter(np.float32(inside[0]), np.float32(1.0))
    sq = oracle.square_distance(q, xyz)[0, 0]
    idx, cnt = oracle.ball_query(0.2, 4, xyz, q, return_counts=True)
    assert cnt[0, 0] == int((~(sq > r2f)).sum())
    assert idx[0, 0, 0] == 0


def _sd(g, prefix):
    return {k[len(prefix) + 3:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(prefix + "sd_")}


def _close(a, b, what, rtol=RTOL, atol=1e-5):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    err = np.abs(a - b).max()
    scale = max(np.abs(b).max(), 1.0)
    assert err <= atol + rtol * scale, f"{what}: max err {err:.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("sa,xyzk,featk,cfg", [
    ("sa1", "xyz", None, dict(npoint=128, radius=0.2, nsample=32)),
    ("sa2", "xyz2", "feats2", dict(npoint=64, radius=0.4, nsample=64)),
    ("sa3", "xyz3", "feats3", dict(npoint=None, radius=None, nsample=None, group_all=True)),
])
@pytest.mark.parametrize("train", [False, True])
def test_set_abstraction_matches_reference(golden, sa, xyzk, featk, cfg, train):
    g = golden("g3_sa")
    tag = f"{sa}_{'train' if train else 'eval'}"
    sd = _sd(g, sa + "_")
    layers = T.layers_from_state(sd, "")
    for L in layers:
        for k in ("weight", "bias", "gamma", "beta"):
            L[k] = L[k].clone().requires_grad_(True)
    xyz = torch.from_numpy(g[xyzk])
    feats = None if featk is None else torch.from_numpy(g[featk]).requires_grad_(True)
    start = g[tag + "_fps_start"] if (tag + "_fps_start") in g.files else None
    new_xyz, new_feats = T.set_abstraction(xyz, feats, layers, fps_start=start, train=train, **cfg)
    _close(new_xyz.permute(0, 2, 1).detach(), g[tag + "_new_xyz"], "new_xyz", atol=0)
    _close(new_feats.permute(0, 2, 1).detach(), g[tag + "_new_points"], "new_points")
    gout = torch.from_numpy(g[tag + "_gout"]).permute(0, 2, 1)
    (new_feats * gout).sum().backward()
    for i, L in enumerate(layers):
        _close(L["weight"].grad[:, :, None, None], g[f"{tag}_grad_mlp_convs.{i}.weight"], f"dW{i}", atol=2e-4, rtol=1e-4)
        _close(L["gamma"].grad, g[f"{tag}_grad_mlp_bns.{i}.weight"], f"dgamma{i}", atol=2e-4, rtol=1e-4)
        _close(L["beta"].grad, g[f"{tag}_grad_mlp_bns.{i}.bias"], f"dbeta{i}", atol=2e-4, rtol=1e-4)
        if not train:
            _close(L["bias"].grad, g[f"{tag}_grad_mlp_convs.{i}.bias"], f"dbias{i}", atol=2e-4, rtol=1e-4)
    if feats is not None:
        _close(feats.grad.permute(0, 2, 1), g[tag + "_grad_feats"], "dfeats", atol=2e-4, rtol=1e-4)
    if train:
        for i, L in enumerate(layers):
            _close(L["running_mean"], g[f"{tag}_after_mlp_bns.{i}.running_mean"], "running_mean")
            _close(L["running_var"], g[f"{tag}_after_mlp_bns.{i}.running_var"], "running_var")


@pytest.mark.parametrize("train", [False, True])
def test_msg_matches_reference(golden, train):
    g = golden("g4_msg")
    tag = "train" if train else "eval"
    sd = _sd(g, "msg_")
    blocks = [T.layers_from_state(sd, "", convs=f"conv_blocks.{i}", bns=f"bn_blocks.{i}") for i in range(3)]
    feats = torch.from_numpy(g["feats"]).requires_grad_(True)
    new_xyz, out = T.set_abstraction_msg(torch.from_numpy(g["xyz"]), feats, blocks, 64, [0.1, 0.2, 0.4], [8, 16, 32],
                                         g["fps_start"], train)
    _close(new_xyz.permute(0, 2, 1), g[tag + "_new_xyz"], "new_xyz", atol=0)
    _close(out.permute(0, 2, 1).detach(), g[tag + "_new_points"], "new_points")
    (out * torch.from_numpy(g[tag + "_gout"]).permute(0, 2, 1)).sum().backward()
    _close(feats.grad.permute(0, 2, 1), g[tag + "_grad_feats"], "dfeats", atol=2e-4, rtol=1e-4)


def test_full_model_eval_matches_reference(golden):
    g = golden("g5_model")
    sd = _sd(g, "")
    out, sm_out, mask_conf = T.strokemasks_forward(sd, torch.from_numpy(g["xyz"]), [g["fps_start1"], g["fps_start2"]],
                                                   train=False, out_vectors=99, n_masks=6)
    _close(out, g["out"], "out")
    _close(sm_out, g["sm_out"], "sm_out")
    _close(mask_conf, g["mask_conf"], "mask_conf")


def test_chamfer_wrapper_matches_reference(golden):
    """Reference wrapper logic (padding, reductions, direction select) around the contract-derived knn."""
    g = golden("g6_cham")
    y_pred, traj, pc = (torch.from_numpy(g[k]) for k in ("y_pred", "traj", "traj_as_pc"))
    B = y_pred.shape[0]
    assert np.array_equal(T.O.padded_lengths(g["traj"]), g["n_seg"])
    assert np.array_equal(T.O.padded_lengths(g["traj_as_pc"]), g["n_pts"])
    calls = {
        "c1": (y_pred, traj, dict(padded=True, asymmetric=True, return_matching=True, point_reduction=None, batch_reduction=None)),
        "c2": (y_pred.reshape(B, -1, 6), pc, dict(padded=True, reverse_asymmetric=True)),
        "c3": (y_pred, traj, dict(padded=True, reverse_asymmetric=True)),
        "c4": (y_pred.reshape(B, -1, 6), pc, dict(padded=True)),
        "c5": (torch.from_numpy(g["xs"]), torch.from_numpy(g["ys"]), dict()),
        "c6": (torch.from_numpy(g["xs"]), torch.from_numpy(g["ys"]), dict(batch_reduction="sum", point_reduction="sum")),
        "c7": (torch.from_numpy(g["xs"]), torch.from_numpy(g["ys"]), dict(batch_reduction=None, point_reduction="mean")),
    }
    for tag, (x, y, kw) in calls.items():
        x = x.clone().requires_grad_(True)
        res = T.chamfer_distance(x, y, **kw)
        d = res[0] if isinstance(res, tuple) else res
        _close(d.detach(), g[tag + "_dist"], tag + " dist", rtol=1e-6, atol=1e-6)
        w = torch.from_numpy(g[tag + "_w"]) if (tag + "_w") in g.files else None
        ((d * w).sum() if w is not None else d).backward()
        _close(x.grad, g[tag + "_gx"], tag + " grad", rtol=1e-5, atol=1e-6)
        if isinstance(res, tuple):
            assert np.array_equal(res[1].numpy(), g[tag + "_idx_x"])
            assert np.array_equal(res[2].numpy(), g[tag + "_idx_y"])


@pytest.mark.parametrize("tag", ["cub", "win"])
def test_mask_loss_matches_reference(golden, tag):
    g = golden("g7_mask")
    cfg = dict(weight_asymm_segment_chamfer=1.0, weight_reverse_asymm_point_chamfer=100, weight_reverse_asymm_segment_chamfer=0.01,
               explicit_weight_stroke_masks=1.0, explicit_weight_stroke_masks_confidence=100.0,
               explicit_no_stroke_weight=float(g[tag + "_no_stroke_weight"]))
    yp = torch.from_numpy(g[tag + "_y_pred"]).requires_grad_(True)
    mk = torch.from_numpy(g[tag + "_masks"]).requires_grad_(True)
    sc = torch.from_numpy(g[tag + "_scores"]).requires_grad_(True)
    loss = T.asymm_v6_loss(yp, torch.from_numpy(g[tag + "_traj"]), mk, sc, torch.from_numpy(g[tag + "_stroke_ids"]),
                           torch.from_numpy(g[tag + "_traj_as_pc"]), cfg)
    _close(loss.detach(), g[tag + "_loss"], "loss", rtol=1e-5)
    loss.backward()
    _close(yp.grad, g[tag + "_g_y_pred"], "g_y_pred", rtol=1e-4, atol=1e-6)
    _close(mk.grad, g[tag + "_g_masks"], "g_masks", rtol=1e-5, atol=1e-6)
    _close(sc.grad, g[tag + "_g_scores"], "g_scores", rtol=1e-5, atol=1e-6)
    mk2 = torch.from_numpy(g[tag + "_masks"]).requires_grad_(True)
    sc2 = torch.from_numpy(g[tag + "_scores"]).requires_grad_(True)
    ml = T.stroke_masks_loss(torch.from_numpy(g[tag + "_idx_x"]), mk2, sc2, torch.from_numpy(g[tag + "_stroke_ids"]),
                             1.0, 100.0, cfg["explicit_no_stroke_weight"])
    _close(ml.detach(), g[tag + "_mask_loss"], "mask_loss", rtol=1e-5)


def test_hungarian_matches_reference(golden):
    g = golden("g8_hung")
    out = torch.from_numpy(g["outputs"])
    tg = [torch.from_numpy(g[f"target{b}"]) for b in range(3)]
    for b, (i, j) in enumerate(T.hungarian_match(out, tg)):
        assert np.array_equal(i, g[f"i{b}"]) and np.array_equal(j, g[f"j{b}"]), b


def test_lsap_matches_scipy(oracle):
    """scipy.optimize.linear_sum_assignment is the reference's own solver (loss_handler.py:875)."""
    from scipy.optimize import linear_sum_assignment as lsa
    rng = np.random.default_rng(0)
    shapes = [(1, 1), (1, 5), (5, 1), (6, 5), (5, 6), (22, 21), (41, 40), (40, 41), (64, 64), (130, 97)]
    for shape in shapes:
        for kind in ("normal", "ties"):
            c = rng.normal(size=shape) if kind == "normal" else rng.integers(0, 3, size=shape).astype(float)
            a, b = oracle.linear_sum_assignment(c)
            r, cc = lsa(c)
            assert np.array_equal(a, r) and np.array_equal(b, cc), (shape, kind)
    a, b = oracle.linear_sum_assignment(np.ones((7, 7)))
    assert np.array_equal(b, np.arange(7))


def test_three_nn_interpolation_bit_exact(golden, oracle):
    """PointNetFeaturePropagation's interpolation (models/pointnet2_utils.py:310-317): indices, distances, weights and the
    interpolated features reproduce the reference bit for bit -- including the coincident points of an FPS subset, whose
    expanded-form distance is 0 +- rounding noise (negative for some), i.e. huge / negative reciprocals."""
    g = golden("g10_fp")
    for t in "ab":
        d, i, w = oracle.three_nn(g[t + "_xyz1"], g[t + "_xyz2"])
        assert np.array_equal(i, g[t + "_nn_idx"]), t
        assert np.array_equal(d.view(np.int32), g[t + "_nn_dist"].view(np.int32)), t
        assert np.array_equal(w.view(np.int32), g[t + "_nn_weight"].view(np.int32)), t
        p2 = np.ascontiguousarray(g[t + "_points2"].transpose(0, 2, 1))
        out = oracle.three_interpolate(p2, i, w)
        assert np.array_equal(out.view(np.int32), g[t + "_interp"].view(np.int32)), t
    assert (g["a_nn_dist"] < 0).any()   # the fixture does contain the negative-distance case


def test_three_interpolate_backward_is_the_adjoint(oracle):
    rng = np.random.default_rng(5)
    B, N, S, D = 2, 200, 40, 7
    xyz1 = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    xyz2 = rng.uniform(-1, 1, (B, S, 3)).astype(np.float32)
    _, idx, w = oracle.three_nn(xyz1, xyz2)
    p2 = rng.normal(size=(B, S, D)).astype(np.float32)
    go = rng.normal(size=(B, N, D)).astype(np.float32)
    lhs = (oracle.three_interpolate(p2, idx, w).astype(np.float64) * go).sum()
    rhs = (oracle.three_interpolate_bwd(go, idx, w, S).astype(np.float64) * p2).sum()
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))



def test_full_model_train_mode_matches_reference(golden):
    """g15: the reference model of g5 in train mode (dropout p = 0) on 8 clouds -- the oracle's train-mode restatement
    (batch statistics in every BatchNorm, running-stat updates) against the imported reference's outputs."""
    g5, g = golden("g5_model"), golden("g15_train")
    sd = _sd(g5, "")
    out, sm_out, mask_conf = T.strokemasks_forward(sd, torch.from_numpy(g["xyz"]), [g["fps_start1"], g["fps_start2"]],
                                                   train=True, out_vectors=99, n_masks=6)
    _close(out, g["out"], "out", rtol=1e-4, atol=1e-5)          # BatchNorm1d over 8 rows amplifies rounding ~10x
    _close(sm_out, g["sm_out"], "sm_out", rtol=1e-4, atol=1e-5)
    _close(mask_conf, g["mask_conf"], "mask_conf", rtol=1e-4, atol=1e-5)
    for k in ("sa1.mlp_bns.0.running_mean", "sa2.mlp_bns.2.running_var", "sa3.mlp_bns.1.running_mean", "bn1.running_var"):
        _close(sd[k], g["after_" + k], k, rtol=2e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------ g17: siblings, returnfps
SIBLINGS = {
    "sops": (lambda pc: pc.PointNet2Regressor_SoPs(out_vectors=7, outdim=3, outdim_orient=3, weight_orient=0.25, hidden_size=(64, 64),
                                                   sop_confidence_scores=True), 171, dict(out_vectors=7)),
    "bbox": (lambda pc: pc.PointNet2Regressor_3Dbbox(out_bboxes=5, hidden_size=(64, 64)), 172, dict(out_vectors=5)),
    "sw": (lambda pc: pc.PointNet2Regressor_StrokeWise(outdim=12, outdim_orient=12, weight_orient=0.25, out_vectors=9, hidden_size=(64, 64),
                                                       stroke_confidence_scores=True, point_confidence_scores=True,
                                                       n_points_per_out_vector=4), 173, dict(out_vectors=9, n_points=4)),
}


def seeded_module(ctor, seed, g, prefix):
    """g9 / g17 store per-tensor checksums of the reference module's weights; the same seed and registration order reproduce them."""
    torch.manual_seed(seed)
    m = ctor()
    sd = m.state_dict()
    keys = [k[len(prefix):] for k in g.files if k.startswith(prefix)]
    assert list(sd.keys()) == keys          # same keys, same order as the reference module
    for k in keys:
        assert abs(float(sd[k].double().abs().sum()) - float(g[prefix + k])) <= 1e-9 * max(1.0, float(g[prefix + k])), k
    return m


@pytest.mark.parametrize("tag", sorted(SIBLINGS))
def test_sibling_regressors_match_reference(golden, tag):
    """g17: PointNet2Regressor_SoPs / _3Dbbox / _StrokeWise (models/pointnet2_cls_ssg.py:85, 177, 463): the oracle's restatement
    against the imported reference, eval mode and train mode (dropout p = 0)."""
    from maskplanner_amd import pointnet2_cls_ssg as pc
    g = golden("g17_siblings")
    ctor, seed, kw = SIBLINGS[tag]
    m = seeded_module(lambda: ctor(pc), seed, g, tag + "_ck_")
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    xyz = torch.from_numpy(g["xyz"])
    outs = T.sibling_forward(tag, sd, xyz, [g[tag + "_eval_s1"], g[tag + "_eval_s2"]], False, **kw)
    for i, o in enumerate(outs):
        _close(o, g[f"{tag}_eval_out{i}"], f"{tag} eval out{i}", rtol=1e-5, atol=1e-5)
    outs = T.sibling_forward(tag, sd, xyz, [g[tag + "_train_s1"], g[tag + "_train_s2"]], True, **kw)
    for i, o in enumerate(outs):
        _close(o, g[f"{tag}_train_out{i}"], f"{tag} train out{i}", rtol=2e-4, atol=1e-5)   # BatchNorm1d over 4 rows
    for k in ("sa1.mlp_bns.0.running_mean", "sa2.mlp_bns.2.running_var", "bn2.running_mean"):
        _close(sd[k], g[f"{tag}_after_{k}"], k, rtol=2e-5, atol=1e-6)


def test_sample_and_group_returnfps_matches_reference(golden, oracle):
    """g17: the two extra return values of sample_and_group(returnfps=True) (models/pointnet2_utils.py:144-145)."""
    g = golden("g17_siblings")
    xyz, feats = g["xyz"], g["rf_feats"]
    fidx = oracle.fps(xyz, 64, g["rf_start"])
    assert np.array_equal(fidx, g["rf_fps_idx"])
    new_xyz = oracle.index_points(xyz, fidx)
    assert np.array_equal(new_xyz, g["rf_new_xyz"])
    idx = oracle.ball_query(0.3, 16, xyz, new_xyz)
    assert np.array_equal(oracle.index_points(xyz, idx.reshape(idx.shape[0], -1)).reshape(g["rf_grouped_xyz"].shape), g["rf_grouped_xyz"])
    grouped = oracle.group(xyz, feats, new_xyz, idx)
    assert np.array_equal(grouped, g["rf_new_points"])
