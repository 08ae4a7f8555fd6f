"""Generate tests/golden/*.npz by running the imported reference on CPU -- TEST INFRASTRUCTURE ONLY.

Run in the build container (reference at /root/reference):   python -m oracle.gen_golden
The fixtures hold inputs and the reference's outputs only (data, no reference source).  The oracle
(mp_oracle.c) and the HIP kernels are both checked against them.

Fixture  producer (reference file:line)                                   pins
g1_fps   models/pointnet2_utils.py:65-86   farthest_point_sample           FPS indices, bit-exact
g2_bq    models/pointnet2_utils.py:89-109  query_ball_point                ball-query indices, bit-exact
g2_sqd   models/pointnet2_utils.py:21-42   square_distance                 expanded-form rounding
g3_sa    models/pointnet2_utils.py:171-216 PointNetSetAbstraction          SA block fwd/bwd (train+eval)
g4_msg   models/pointnet2_utils.py:219-276 PointNetSetAbstractionMsg       MSG block fwd/bwd
g5_model models/pointnet2_cls_ssg.py:233-344 PointNet2Regressor_StrokeMasks full forward (eval)
g6_cham  pytorch3d_chamfer.py:76-344 chamfer_distance (pytorch3d stand-in = oracle knn: contract-derived)
g7_mask  loss_handler.py:596-666,816-935  asymm_v6 loss + stroke-mask loss
g8_hung  models/hungarianMatcher.py:31-63 HungarianMatcher
g9_seg   models/pointnet2_seg.py:14-96,258-339 PointNet2Segmenter_v1 / _PaintNet_v1 (eval forward)
g10_fp   models/pointnet2_utils.py:279-329 PointNetFeaturePropagation      3-NN interpolation + MLP fwd/bwd (train+eval)
                                         (its two `*_g_points2` arrays come out of torch's multi-threaded CPU scatter-add: they
                                         regenerate equal to ~1e-6, not bit for bit; every other fixture regenerates byte-identical)
g11_smooth loss_handler.py:830,841-844,959-964 smooth_target_stroke_masks  MSE mask matching + loss on g7's inputs
g12_flags pytorch3d_chamfer.py:180-291  chamfer_distance: velocities / min_centroids / avoid_in_sequence_collapsing /
                                         soft_attraction / normals / weights (values + gradients)
g13_losses loss_handler.py:521-552,566-593,669-801,990-1009 + metrics_handler.py:285-308: the sibling loss terms
                                         (asymm_v11, symm_v1, chamfer_with_stroke_masks, chamfer, symm_point, attraction, emd,
                                         per_segment_confidence) and stroke_masks_metrics
g14_collate utils/dataset/paintnet_ODv1.py:726-847 Paintnet_ODv1_CollateBatch.__call__ on ragged synthetic samples
g16_lambda utils/pointcloud.py:294-413 get_sequences_of_lambda_points (+ add_padding) on ragged synthetic strokes
g15_train models/pointnet2_cls_ssg.py:233-344 the full model of g5 (same weights) in TRAIN mode (dropout p = 0) on 8 clouds:
                                         outputs, running statistics after the pass, gradients of a linear functional
g17_siblings models/pointnet2_cls_ssg.py:85,177,463 PointNet2Regressor_SoPs / _3Dbbox / _StrokeWise (eval + train, gradients);
                                         models/pointnet2_seg.py:14-96,258-339 the segmenters' backward;
                                         models/pointnet2_utils.py:144-145 sample_and_group(returnfps=True)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import as R  # noqa: E402
from maskplanner_amd import synthetic as syn  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def ref_fps(pu, xyz, npoint, seed):
    """Run the reference FPS; recover its random start by replaying the CPU generator."""
    B, N, _ = xyz.shape
    torch.manual_seed(seed)
    start = torch.randint(0, N, (B,), dtype=torch.long)
    torch.manual_seed(seed)
    idx = pu.farthest_point_sample(torch.from_numpy(xyz), npoint)
    assert (idx[:, 0] == start).all()
    return start.numpy(), idx.numpy()


def g1_fps(pu):
    print("g1_fps")
    rng = np.random.default_rng(101)
    cases = {}
    clouds = {
        "ucube5120": (syn.point_cloud(rng, 2, 5120, "ucube"), 512),
        "cuboid5120": (syn.point_cloud(rng, 2, 5120, "cuboid"), 512),
        "cuboid512": (syn.point_cloud(rng, 3, 512, "cuboid"), 128),
        "ragged777": (syn.point_cloud(rng, 2, 777, "ucube"), 100),
    }
    # duplicates (coincident points) and exact ties: a regular integer lattice has many equal distances
    lat = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(-1, 3)
    lat = (lat.astype(np.float32) - 3.5) / 4.0
    dup = np.concatenate([lat, lat[:64]], 0)
    clouds["lattice_dup"] = (np.stack([dup, dup[::-1].copy()]), 96)
    for i, (k, (xyz, npoint)) in enumerate(clouds.items()):
        start, idx = ref_fps(pu, xyz, npoint, 500 + i)
        cases[k + "_xyz"] = xyz
        cases[k + "_start"] = start
        cases[k + "_idx"] = idx
    save("g1_fps", **cases)


def g2_bq(pu):
    print("g2_bq")
    rng = np.random.default_rng(202)
    cases = {}

    def add(name, xyz, npoint, radius, K, seed):
        start, fidx = ref_fps(pu, xyz, npoint, seed)
        new_xyz = pu.index_points(torch.from_numpy(xyz), torch.from_numpy(fidx)).numpy()
        idx = pu.query_ball_point(radius, K, torch.from_numpy(xyz), torch.from_numpy(new_xyz)).numpy()
        cases[name + "_xyz"] = xyz
        cases[name + "_new_xyz"] = new_xyz
        cases[name + "_radius"] = np.float64(radius)
        cases[name + "_K"] = np.int64(K)
        cases[name + "_idx"] = idx.astype(np.int32)  # values < 2^31; stored narrow to keep the fixture small

    add("sa1_ucube", syn.point_cloud(rng, 2, 5120, "ucube"), 512, 0.2, 32, 11)
    add("sa1_cuboid", syn.point_cloud(rng, 2, 5120, "cuboid"), 512, 0.2, 32, 12)
    add("sa2_cuboid", syn.point_cloud(rng, 3, 512, "cuboid"), 128, 0.4, 64, 13)
    add("dbg1024", syn.point_cloud(rng, 2, 1024, "cuboid"), 512, 0.2, 32, 14)
    add("ragged", syn.point_cloud(rng, 2, 777, "ucube"), 100, 0.35, 48, 15)

    # engineered: points within a few ulp of the sphere r around each query (threshold rounding)
    for radius, tag in ((0.2, "r02"), (0.4, "r04")):
        q = rng.uniform(-0.5, 0.5, size=(1, 16, 3)).astype(np.float32)
        pts = []
        for s in range(16):
            d = rng.normal(size=(96, 3))
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            eps = rng.integers(-6, 7, size=(96, 1)) * 2.0 ** -24
            pts.append((q[0, s].astype(np.float64) + d * radius * (1.0 + eps)).astype(np.float32))
        xyz = np.concatenate(pts)[None]  # [1,1536,3]
        perm = rng.permutation(xyz.shape[1])
        xyz = np.ascontiguousarray(xyz[:, perm])
        sq = pu.square_distance(torch.from_numpy(q), torch.from_numpy(xyz)).numpy()
        idx = pu.query_ball_point(radius, 64, torch.from_numpy(xyz), torch.from_numpy(q)).numpy()
        r2f = np.float32(radius ** 2)
        near = np.abs(sq - r2f) <= 4 * np.spacing(r2f)
        print(f"   threshold case {tag}: {near.sum()} pairs within 4 ulp of r^2, {(sq == r2f).sum()} exactly equal")
        cases[f"thr_{tag}_xyz"] = xyz
        cases[f"thr_{tag}_new_xyz"] = q
        cases[f"thr_{tag}_radius"] = np.float64(radius)
        cases[f"thr_{tag}_K"] = np.int64(64)
        cases[f"thr_{tag}_idx"] = idx.astype(np.int32)
        cases[f"thr_{tag}_sqd"] = sq
    save("g2_bq", **cases)

    # expanded-form rounding on the real shapes (values, not only memberships)
    xyz = syn.point_cloud(rng, 1, 5120, "cuboid")
    q = xyz[:, rng.permutation(5120)[:16]]
    sq = pu.square_distance(torch.from_numpy(q), torch.from_numpy(xyz)).numpy()
    xyz2 = syn.point_cloud(rng, 2, 512, "ucube")
    q2 = xyz2[:, :48]
    sq2 = pu.square_distance(torch.from_numpy(q2), torch.from_numpy(xyz2)).numpy()
    save("g2_sqd", a_src=q, a_dst=xyz, a_out=sq, b_src=q2, b_dst=xyz2, b_out=sq2)


def _seed_module(mod, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in mod.named_parameters():
            if p.ndim > 1:
                fan_in = p[0].numel()
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) / fan_in ** 0.5)
            elif name.endswith("weight"):  # BN gamma (some negative, to exercise the min-pool branch)
                p.copy_(torch.rand(p.shape, generator=g) * 1.5 - 0.25)
            else:
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.1)


def _state(mod):
    return {"sd_" + k: v.detach().numpy().copy() for k, v in mod.state_dict().items()}


def g3_sa(pu):
    print("g3_sa")
    rng = np.random.default_rng(303)
    cases = {}

    def run(tag, sa, xyz, feats, fps_seed, train):
        sa.train(train)
        x = torch.from_numpy(xyz).permute(0, 2, 1).contiguous().requires_grad_(True)
        f = None if feats is None else torch.from_numpy(feats).permute(0, 2, 1).contiguous().requires_grad_(True)
        if not sa.group_all:
            start, _ = ref_fps(pu, xyz, sa.npoint, fps_seed)
            cases[tag + "_fps_start"] = start
        torch.manual_seed(fps_seed)
        new_xyz, new_points = sa(x, f)
        gw = torch.from_numpy(rng.normal(size=tuple(new_points.shape)).astype(np.float32))
        cases[tag + "_gout"] = gw.numpy()
        params = list(sa.parameters())
        grads = torch.autograd.grad((new_points * gw).sum(), params + ([f] if f is not None else []), allow_unused=True)
        cases[tag + "_new_xyz"] = new_xyz.detach().numpy()
        cases[tag + "_new_points"] = new_points.detach().numpy()
        for (n, _), g in zip(sa.named_parameters(), grads):
            cases[tag + "_grad_" + n] = g.numpy()
        if f is not None:
            cases[tag + "_grad_feats"] = grads[-1].numpy()
        if train:
            for k, v in sa.state_dict().items():
                if "running" in k:
                    cases[tag + "_after_" + k] = v.numpy().copy()

    # SA1-like (no features), SA2-like (features), group_all -- small channel counts, real structure
    xyz = syn.point_cloud(rng, 2, 1024, "cuboid")
    cases["xyz"] = xyz
    sa1 = pu.PointNetSetAbstraction(npoint=128, radius=0.2, nsample=32, in_channel=3, mlp=[32, 32, 64], group_all=False)
    _seed_module(sa1, 1)
    cases.update({"sa1_" + k: v for k, v in _state(sa1).items()})
    run("sa1_eval", sa1, xyz, None, 21, False)
    run("sa1_train", sa1, xyz, None, 21, True)

    xyz2 = syn.point_cloud(rng, 2, 256, "cuboid")
    feats2 = rng.normal(size=(2, 256, 64)).astype(np.float32)
    cases["xyz2"] = xyz2
    cases["feats2"] = feats2
    sa2 = pu.PointNetSetAbstraction(npoint=64, radius=0.4, nsample=64, in_channel=64 + 3, mlp=[64, 64, 128], group_all=False)
    _seed_module(sa2, 2)
    cases.update({"sa2_" + k: v for k, v in _state(sa2).items()})
    run("sa2_eval", sa2, xyz2, feats2, 22, False)
    run("sa2_train", sa2, xyz2, feats2, 22, True)

    xyz3 = syn.point_cloud(rng, 4, 128, "cuboid")
    feats3 = rng.normal(size=(4, 128, 61)).astype(np.float32)
    cases["xyz3"] = xyz3
    cases["feats3"] = feats3
    sa3 = pu.PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=61 + 3, mlp=[64, 96, 160], group_all=True)
    _seed_module(sa3, 3)
    cases.update({"sa3_" + k: v for k, v in _state(sa3).items()})
    run("sa3_eval", sa3, xyz3, feats3, 23, False)
    run("sa3_train", sa3, xyz3, feats3, 23, True)
    save("g3_sa", **cases)


def g4_msg(pu):
    print("g4_msg")
    rng = np.random.default_rng(404)
    cases = {}
    xyz = syn.point_cloud(rng, 2, 512, "cuboid")
    feats = rng.normal(size=(2, 512, 13)).astype(np.float32)
    msg = pu.PointNetSetAbstractionMsg(64, [0.1, 0.2, 0.4], [8, 16, 32], 13, [[16, 16, 32], [16, 24, 32], [16, 24, 48]])
    _seed_module(msg, 4)
    cases.update({"msg_" + k: v for k, v in _state(msg).items()})
    cases["xyz"] = xyz
    cases["feats"] = feats
    for train in (False, True):
        tag = "train" if train else "eval"
        msg.train(train)
        start, _ = ref_fps(pu, xyz, 64, 31)
        cases["fps_start"] = start
        x = torch.from_numpy(xyz).permute(0, 2, 1).contiguous()
        f = torch.from_numpy(feats).permute(0, 2, 1).contiguous().requires_grad_(True)
        torch.manual_seed(31)
        new_xyz, new_points = msg(x, f)
        gw = torch.from_numpy(rng.normal(size=tuple(new_points.shape)).astype(np.float32))
        grads = torch.autograd.grad((new_points * gw).sum(), list(msg.parameters()) + [f])
        cases[tag + "_gout"] = gw.numpy()
        cases[tag + "_new_xyz"] = new_xyz.detach().numpy()
        cases[tag + "_new_points"] = new_points.detach().numpy()
        for (n, _), g in zip(msg.named_parameters(), grads):
            cases[tag + "_grad_" + n] = g.numpy()
        cases[tag + "_grad_feats"] = grads[-1].numpy()
    save("g4_msg", **cases)


def g5_model():
    print("g5_model")
    pc = R.pointnet2_cls_ssg()
    pu = R.pointnet2_utils()
    rng = np.random.default_rng(505)
    torch.manual_seed(5)
    # cuboids debug shape, reduced head width/out_vectors to keep the fixture small (structure unchanged)
    model = pc.PointNet2Regressor_StrokeMasks(outdim=12, outdim_orient=12, weight_orient=0.25, out_vectors=99,
                                              hidden_size=(64, 64), pred_stroke_masks=True, n_stroke_masks=6,
                                              mask_confidence_scores=True, segment_confidence_scores=False)
    model.eval()
    xyz = syn.point_cloud(rng, 2, 1024, "cuboid")
    s1, _ = ref_fps(pu, xyz, 512, 41)  # SA1 draw
    torch.manual_seed(41)
    _ = torch.randint(0, 1024, (2,))
    s2 = torch.randint(0, 512, (2,)).numpy()  # SA2 draw follows in the same generator stream
    torch.manual_seed(41)
    with torch.no_grad():
        out, sm_out, mask_conf, seg_conf = model(torch.from_numpy(xyz).permute(0, 2, 1))
    assert seg_conf is None
    # only the encoder weights are big-ish; heads are small here
    sd = {"sd_" + k: v.numpy().copy() for k, v in model.state_dict().items()}
    save("g5_model", xyz=xyz, fps_start1=s1, fps_start2=s2, out=out.numpy(), sm_out=sm_out.numpy(),
         mask_conf=mask_conf.numpy(), **sd)


def g6_cham():
    print("g6_cham")
    ch = R.chamfer_module()
    rng = np.random.default_rng(606)
    cases = {}
    B, S = 3, 99
    traj, traj_as_pc, stroke_ids, n_seg, n_pts = syn.ground_truth(rng, B, syn.Category("t", S, 6, 3, 6, 120, 300))
    y_pred = rng.uniform(-1, 1, size=(B, S, 24)).astype(np.float32)
    cases.update(y_pred=y_pred, traj=traj, traj_as_pc=traj_as_pc, n_seg=n_seg, n_pts=n_pts)

    def call(tag, x, y, **kw):
        xt = torch.from_numpy(x).requires_grad_(True)
        yt = torch.from_numpy(y.copy())
        res = ch.chamfer_distance(xt, yt, **kw)
        d = res[0]
        if d.ndim == 0:
            (gx,) = torch.autograd.grad(d, xt)
        else:
            w = torch.from_numpy(rng.normal(size=tuple(d.shape)).astype(np.float32))
            cases[tag + "_w"] = w.numpy()
            (gx,) = torch.autograd.grad((d * w).sum(), xt)
        cases[tag + "_dist"] = d.detach().numpy()
        cases[tag + "_gx"] = gx.numpy()
        if len(res) == 4:
            cases[tag + "_idx_x"] = res[2].numpy()
            cases[tag + "_idx_y"] = res[3].numpy()

    # the three maskplanner call patterns (loss_handler.py:604-611, 633-637, 642-645)
    call("c1", y_pred, traj, padded=True, asymmetric=True, return_matching=True, point_reduction=None, batch_reduction=None)
    call("c2", y_pred.reshape(B, -1, 6), traj_as_pc, padded=True, reverse_asymmetric=True)
    call("c3", y_pred, traj, padded=True, reverse_asymmetric=True)
    # metrics_handler.get_pcd pattern (symmetric, mean/mean, padded) and plain unpadded variants
    call("c4", y_pred.reshape(B, -1, 6), traj_as_pc, padded=True)
    xs = rng.normal(size=(2, 50, 3)).astype(np.float32)
    ys = rng.normal(size=(2, 70, 3)).astype(np.float32)
    cases.update(xs=xs, ys=ys)
    call("c5", xs, ys)
    call("c6", xs, ys, batch_reduction="sum", point_reduction="sum")
    call("c7", xs, ys, batch_reduction=None, point_reduction="mean")
    save("g6_cham", **cases)


def g7_mask():
    print("g7_mask")
    lh = R.loss_handler_module()
    rng = np.random.default_rng(707)
    cases = {}
    for tag, B, S, M, cat in (("cub", 3, 99, 6, syn.Category("t", 99, 6, 6, 6, 150, 300)),
                              ("win", 2, 149, 22, syn.Category("t", 149, 22, 8, 22, 300, 420))):
        traj, traj_as_pc, stroke_ids, n_seg, n_pts = syn.ground_truth(rng, B, cat)
        # predictions near GT segments so that several strokes get matched
        y_pred = np.empty((B, S, 24), dtype=np.float32)
        for b in range(B):
            pick = rng.integers(0, n_seg[b], size=S)
            y_pred[b] = traj[b, pick] + rng.normal(scale=0.02, size=(S, 24)).astype(np.float32)
        masks = rng.normal(size=(B, M, S)).astype(np.float32)
        scores = rng.normal(size=(B, M)).astype(np.float32)
        handler = object.__new__(lh.LossHandler)
        handler.config = R.maskplanner_loss_config(explicit_no_stroke_weight=0.5 if tag == "win" else 1.0)
        yp = torch.from_numpy(y_pred).requires_grad_(True)
        mk = torch.from_numpy(masks).requires_grad_(True)
        sc = torch.from_numpy(scores).requires_grad_(True)
        loss = handler.get_asymm_v6_chamfer_with_stroke_masks(
            y_pred=yp, y=torch.from_numpy(traj), pred_stroke_masks=mk, mask_scores=sc, seg_logits=None,
            stroke_ids=torch.from_numpy(stroke_ids), traj_as_pc=torch.from_numpy(traj_as_pc))
        g = torch.autograd.grad(loss, [yp, mk, sc])
        # the mask term alone (needs idx_x of chamfer call 1)
        ch = R.chamfer_module()
        d1, _, idx_x, _ = ch.chamfer_distance(torch.from_numpy(y_pred), torch.from_numpy(traj), padded=True, asymmetric=True,
                                              return_matching=True, point_reduction=None, batch_reduction=None)
        mk2 = torch.from_numpy(masks).requires_grad_(True)
        sc2 = torch.from_numpy(scores).requires_grad_(True)
        mloss = handler.get_stroke_masks_loss(idx_x, mk2, sc2, torch.from_numpy(stroke_ids), nn_distance=d1, smooth_targets=False)
        gm = torch.autograd.grad(mloss, [mk2, sc2])
        cases.update({
            tag + "_y_pred": y_pred, tag + "_traj": traj, tag + "_traj_as_pc": traj_as_pc, tag + "_stroke_ids": stroke_ids,
            tag + "_masks": masks, tag + "_scores": scores, tag + "_no_stroke_weight": np.float64(handler.config["explicit_no_stroke_weight"]),
            tag + "_loss": loss.detach().numpy(), tag + "_g_y_pred": g[0].numpy(), tag + "_g_masks": g[1].numpy(),
            tag + "_g_scores": g[2].numpy(), tag + "_idx_x": idx_x.numpy(), tag + "_mask_loss": mloss.detach().numpy(),
            tag + "_gm_masks": gm[0].numpy(), tag + "_gm_scores": gm[1].numpy(),
        })
    save("g7_mask", **cases)


def g8_hung():
    print("g8_hung")
    hm = R.hungarian_matcher()
    rng = np.random.default_rng(808)
    S = 200
    sizes = [180, 150, 200]
    out = rng.uniform(-1, 1, size=(3, S, 24)).astype(np.float32)
    tg = [rng.uniform(-1, 1, size=(n, 24)).astype(np.float32) for n in sizes]
    res = hm.HungarianMatcher()(torch.from_numpy(out), [torch.from_numpy(t) for t in tg])
    cases = {"outputs": out}
    for b, (i, j) in enumerate(res):
        cases[f"target{b}"] = tg[b]
        cases[f"i{b}"] = i.numpy()
        cases[f"j{b}"] = j.numpy()
    save("g8_hung", **cases)


def g9_seg():
    """models/pointnet2_seg.py:14-96, 258-339: the two instantiable segmenters (SA stack + per-point Conv1d head)."""
    print("g9_seg")
    sg = R.pointnet2_seg()
    rng = np.random.default_rng(909)
    cases = {}
    B, N = 2, 1024
    # PaintNet_v1: 3-D points in, lambda poses out
    torch.manual_seed(9)
    m = sg.PointNet2Segmenter_PaintNet_v1(inputdim=3, outdim_trasl=3, outdim_orient=3, weight_orient=0.25, lambda_points=2)
    m.eval()
    xyz = syn.point_cloud(rng, B, N, "cuboid")
    torch.manual_seed(51)
    s1 = torch.randint(0, N, (B,)).numpy()
    s2 = torch.randint(0, 512, (B,)).numpy()
    torch.manual_seed(51)
    with torch.no_grad():
        out = m(torch.from_numpy(xyz).permute(0, 2, 1))
    # weights are regenerated from the seed by the test (same module registration order => same RNG stream); the
    # fixture keeps per-tensor checksums instead of 5 MB of state_dict
    cases.update({"pn_ck_" + k: np.float64(v.double().abs().sum()) for k, v in m.state_dict().items()})
    cases.update(pn_xyz=xyz, pn_fps_start1=s1, pn_fps_start2=s2, pn_out=out.numpy())
    # Segmenter_v1 with ball_in_xyz_space: FPS / ball query on segment centroids, full 24-D segments as features
    torch.manual_seed(10)
    m2 = sg.PointNet2Segmenter_v1(outdim=5, input_orient_dim=3, lambda_points=4, ball_in_xyz_space=True)
    m2.eval()
    segs = rng.uniform(-0.5, 0.5, size=(B, 24, N)).astype(np.float32)
    torch.manual_seed(52)
    t1 = torch.randint(0, N, (B,)).numpy()
    t2 = torch.randint(0, 512, (B,)).numpy()
    torch.manual_seed(52)
    with torch.no_grad():
        out2 = m2(torch.from_numpy(segs))
    cases.update({"sg_ck_" + k: np.float64(v.double().abs().sum()) for k, v in m2.state_dict().items()})
    cases.update(sg_in=segs, sg_fps_start1=t1, sg_fps_start2=t2, sg_out=out2.numpy())
    save("g9_seg", **cases)


def g10_fp(pu):
    """models/pointnet2_utils.py:279-329: feature propagation (3-NN inverse-distance interpolation + Conv1d/BN1d MLP).
    xyz2 is an FPS subset of xyz1 (as in a real decoder), so coincident points -- distances of 0 +- rounding noise in the
    expanded form, i.e. huge and possibly negative reciprocals -- are part of the fixture."""
    print("g10_fp")
    rng = np.random.default_rng(1010)
    cases = {}
    specs = {"a": dict(B=2, N=1024, S=256, D1=6, D2=32, mlp=[32, 16]),     # points1 present
             "b": dict(B=2, N=700, S=64, D1=0, D2=24, mlp=[16]),           # points1 = None (fp1 of the v3 decoder)
             "c": dict(B=2, N=300, S=1, D1=4, D2=8, mlp=[8])}              # S == 1: the repeat branch (:307-308)
    for tag, sp in specs.items():
        B, N, S = sp["B"], sp["N"], sp["S"]
        xyz1 = syn.point_cloud(rng, B, N, "cuboid")
        if S > 1:
            _, fidx = ref_fps(pu, xyz1, S, 1000 + S)
            xyz2 = np.take_along_axis(xyz1, fidx[..., None], 1).copy()
        else:
            xyz2 = xyz1[:, :1].copy()
        p1 = rng.normal(size=(B, sp["D1"], N)).astype(np.float32) if sp["D1"] else None
        p2 = rng.normal(size=(B, sp["D2"], S)).astype(np.float32)
        torch.manual_seed(77)
        m = pu.PointNetFeaturePropagation(sp["D1"] + sp["D2"], sp["mlp"])
        with torch.no_grad():   # non-trivial BN affine + running stats
            for bn in m.mlp_bns:
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.uniform_(-0.3, 0.3)
                bn.running_mean.uniform_(-0.2, 0.2)
                bn.running_var.uniform_(0.5, 1.5)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        cases.update({f"{tag}_sd_{k}": v.numpy() for k, v in sd0.items()})
        t = lambda a: None if a is None else torch.from_numpy(a)
        if S > 1:   # the interpolation alone (reference lines 310-317)
            d = pu.square_distance(t(xyz1), t(xyz2))
            ds, di = d.sort(dim=-1)
            ds, di = ds[:, :, :3], di[:, :, :3]
            rc = 1.0 / (ds + 1e-8)
            w = rc / rc.sum(dim=2, keepdim=True)
            interp = torch.sum(pu.index_points(t(p2).permute(0, 2, 1), di) * w.view(B, N, 3, 1), dim=2)
            cases.update({f"{tag}_nn_dist": ds.numpy(), f"{tag}_nn_idx": di.numpy(), f"{tag}_nn_weight": w.numpy(),
                          f"{tag}_interp": interp.numpy()})
        m.eval()
        with torch.no_grad():
            oe = m(t(xyz1).permute(0, 2, 1), t(xyz2).permute(0, 2, 1), t(p1), t(p2))
        m.train()
        p1t = None if p1 is None else t(p1).requires_grad_(True)
        p2t = t(p2).requires_grad_(True)
        ot = m(t(xyz1).permute(0, 2, 1), t(xyz2).permute(0, 2, 1), p1t, p2t)
        go = rng.normal(size=tuple(ot.shape)).astype(np.float32)
        params = list(m.parameters())
        grads = torch.autograd.grad(ot, ([p1t] if p1t is not None else []) + [p2t] + params, t(go))
        k = 0
        if p1t is not None:
            cases[f"{tag}_g_points1"] = grads[0].numpy()
            k = 1
        cases[f"{tag}_g_points2"] = grads[k].numpy()
        for (name, _), g in zip(m.named_parameters(), grads[k + 1:]):
            cases[f"{tag}_g_{name}"] = g.numpy()
        cases.update({f"{tag}_after_{k2}": v.numpy() for k2, v in m.state_dict().items() if "running" in k2})
        cases.update({f"{tag}_xyz1": xyz1, f"{tag}_xyz2": xyz2, f"{tag}_points2": p2, f"{tag}_out_eval": oe.numpy(),
                      f"{tag}_out_train": ot.detach().numpy(), f"{tag}_grad_out": go})
        if p1 is not None:
            cases[f"{tag}_points1"] = p1
    save("g10_fp", **cases)


def g11_smooth():
    """The `smooth_target_stroke_masks` variant of the stroke-mask loss (loss_handler.py:830, 841-844, 889-893, 959-964;
    off in the shipped configs, default.yaml:117) on the inputs of g7_mask: target masks hold f(nn_distance) instead of 1,
    matching cost and loss are MSE, and the gradient reaches nn_distance."""
    print("g11_smooth")
    lh = R.loss_handler_module()
    ch = R.chamfer_module()
    g7 = np.load(os.path.join(OUT, "g7_mask.npz"))
    cases = {}
    for tag in ("cub", "win"):
        y_pred, traj, stroke_ids = g7[tag + "_y_pred"], g7[tag + "_traj"], g7[tag + "_stroke_ids"]
        masks, scores = g7[tag + "_masks"], g7[tag + "_scores"]
        handler = object.__new__(lh.LossHandler)
        handler.config = R.maskplanner_loss_config(explicit_no_stroke_weight=float(g7[tag + "_no_stroke_weight"]),
                                                   smooth_target_stroke_masks=True)
        d1, _, idx_x, _ = ch.chamfer_distance(torch.from_numpy(y_pred), torch.from_numpy(traj), padded=True, asymmetric=True,
                                              return_matching=True, point_reduction=None, batch_reduction=None)
        dl = d1.detach().clone().requires_grad_(True)
        mk = torch.from_numpy(masks).requires_grad_(True)
        sc = torch.from_numpy(scores).requires_grad_(True)
        ml = handler.get_stroke_masks_loss(idx_x, mk, sc, torch.from_numpy(stroke_ids), nn_distance=dl, smooth_targets=True)
        gm = torch.autograd.grad(ml, [mk, sc, dl])
        # the whole asymm_v6 loss with the switch on (gradient through the chamfer distances into y_pred)
        yp = torch.from_numpy(y_pred).requires_grad_(True)
        mk2 = torch.from_numpy(masks).requires_grad_(True)
        sc2 = torch.from_numpy(scores).requires_grad_(True)
        loss = handler.get_asymm_v6_chamfer_with_stroke_masks(
            y_pred=yp, y=torch.from_numpy(traj), pred_stroke_masks=mk2, mask_scores=sc2, seg_logits=None,
            stroke_ids=torch.from_numpy(stroke_ids), traj_as_pc=torch.from_numpy(g7[tag + "_traj_as_pc"]))
        g = torch.autograd.grad(loss, [yp, mk2, sc2])
        cases.update({tag + "_nn_distance": d1.detach().numpy(), tag + "_idx_x": idx_x.numpy(),
                      tag + "_mask_loss": ml.detach().numpy(), tag + "_gm_masks": gm[0].numpy(), tag + "_gm_scores": gm[1].numpy(),
                      tag + "_gm_distance": gm[2].numpy(), tag + "_loss": loss.detach().numpy(), tag + "_g_y_pred": g[0].numpy(),
                      tag + "_g_masks": g[1].numpy(), tag + "_g_scores": g[2].numpy()})
    save("g11_smooth", **cases)



def g12_flags():
    """Every remaining branch of the reference chamfer wrapper (pytorch3d_chamfer.py:180-291), run through the imported
    reference file itself (kNN stand-in = the oracle's, as for g6)."""
    print("g12_flags")
    ch = R.chamfer_module()
    rng = np.random.default_rng(1212)
    cases = {}

    def call(tag, x, y, grad=True, **kw):
        xt = torch.from_numpy(x.copy()).requires_grad_(grad)
        yt = torch.from_numpy(y.copy()).requires_grad_(grad)
        tk = {k: (torch.from_numpy(v.copy()) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
        d, dn = ch.chamfer_distance(xt, yt, **tk)[:2]
        cases[tag + "_dist"] = d.detach().numpy()
        if dn is not None:
            cases[tag + "_normals"] = dn.detach().numpy()
        if grad:
            tot = d.sum() if dn is None else d.sum() + 0.5 * dn.sum()
            gx, gy = torch.autograd.grad(tot, [xt, yt])
            cases[tag + "_gx"], cases[tag + "_gy"] = gx.numpy(), gy.numpy()

    x6 = rng.uniform(0, 1, size=(2, 40, 6)).astype(np.float32)
    y6 = rng.uniform(0, 1, size=(2, 40, 6)).astype(np.float32)
    xs = rng.uniform(0, 1, size=(2, 30, 24)).astype(np.float32)
    ys = rng.uniform(0, 1, size=(2, 30, 24)).astype(np.float32)
    s3 = xs[..., :3].copy()
    e3 = (s3 + 0.01 * rng.uniform(0, 1, size=s3.shape)).astype(np.float32)      # mostly in-sequence nearest neighbours
    e3b = rng.uniform(0, 1, size=s3.shape).astype(np.float32)                    # unrelated: mostly out-of-sequence
    nx = rng.normal(size=(2, 40, 3)).astype(np.float32)
    ny = rng.normal(size=(2, 40, 3)).astype(np.float32)
    w = np.array([0.5, 2.0], dtype=np.float32)
    cases.update(x6=x6, y6=y6, xs=xs, ys=ys, s3=s3, e3=e3, e3b=e3b, nx=nx, ny=ny, w=w)
    call("vel", x6, y6, velocities=True)                                          # :180-199 (the reference's in-place writes
    call("minc", xs, ys, min_centroids=True)                                      #  into fresh tensors keep autograd intact)
    call("attr", s3, e3, avoid_in_sequence_collapsing=True)                       # :201-222
    call("soft", s3, e3b, avoid_in_sequence_collapsing=True, soft_attraction=True, point_reduction=None, batch_reduction=None)
    call("wn", x6[..., :3].copy(), y6[..., :3].copy(), x_normals=nx, y_normals=ny, weights=w)
    call("wn_sum", x6[..., :3].copy(), y6[..., :3].copy(), x_normals=nx, y_normals=ny, weights=w, batch_reduction="sum", point_reduction="sum")
    call("w0", x6, y6, grad=False, weights=np.zeros(2, dtype=np.float32))         # :160-175 early return
    save("g12_flags", **cases)


def g13_losses():
    """The loss terms of the maskplanner path that g7 / g11 do not reach, through the imported reference LossHandler."""
    print("g13_losses")
    lh = R.loss_handler_module()
    rng = np.random.default_rng(1313)
    B, S, M = 3, 99, 6
    cat = syn.Category("t", S, M, 6, 6, 150, 300)
    traj, traj_as_pc, stroke_ids, n_seg, n_pts = syn.ground_truth(rng, B, cat)
    y_pred = np.empty((B, S, 24), dtype=np.float32)
    for b in range(B):
        pick = rng.integers(0, n_seg[b], size=S)
        y_pred[b] = traj[b, pick] + rng.normal(scale=0.02, size=(S, 24)).astype(np.float32)
    masks = rng.normal(size=(B, M, S)).astype(np.float32)
    scores = rng.normal(size=(B, M)).astype(np.float32)
    seg_logits = rng.uniform(0, 1, size=(B, S)).astype(np.float32)
    cases = dict(y_pred=y_pred, traj=traj, traj_as_pc=traj_as_pc, stroke_ids=stroke_ids, masks=masks, scores=scores,
                 seg_logits=seg_logits)
    cfg_extra = dict(weight_symm_segment_chamfer=0.7, weight_symm_point_chamfer=30.0, soft_attraction=False)

    def run(tag, method, per_segment_confidence=False, wants=("y_pred", "masks", "scores")):
        handler = object.__new__(lh.LossHandler)
        handler.config = R.maskplanner_loss_config(per_segment_confidence=per_segment_confidence, **cfg_extra)
        if method == "get_emd":
            handler.matcher = R.hungarian_matcher().HungarianMatcher()
        t = dict(y_pred=torch.from_numpy(y_pred).requires_grad_(True), masks=torch.from_numpy(masks).requires_grad_(True),
                 scores=torch.from_numpy(scores).requires_grad_(True), seg=torch.from_numpy(seg_logits).requires_grad_(True))
        loss = getattr(handler, method)(y_pred=t["y_pred"], y=torch.from_numpy(traj), pred_stroke_masks=t["masks"],
                                         mask_scores=t["scores"], seg_logits=t["seg"] if per_segment_confidence else None,
                                         stroke_ids=torch.from_numpy(stroke_ids), traj_as_pc=torch.from_numpy(traj_as_pc))
        cases[tag + "_loss"] = loss.detach().numpy()
        names = list(wants) + (["seg"] if per_segment_confidence else [])
        grads = torch.autograd.grad(loss, [t[n] for n in names], allow_unused=True)
        for n, g in zip(names, grads):
            cases[f"{tag}_g_{n}"] = np.zeros(tuple(t[n].shape), np.float32) if g is None else g.numpy()

    run("v11", "get_asymm_v11_chamfer_with_stroke_masks")                        # :669-730
    run("v11c", "get_asymm_v11_chamfer_with_stroke_masks", per_segment_confidence=True)
    run("v6c", "get_asymm_v6_chamfer_with_stroke_masks", per_segment_confidence=True)   # :596-666 with :566-593
    run("symm", "get_symm_v1_chamfer_with_stroke_masks")                         # :733-777
    run("cwm", "get_chamfer_with_stroke_masks")                                  # :780-801
    run("chamfer", "get_chamfer", wants=("y_pred",))                             # :534-552
    run("sympt", "get_symm_point_chamfer", wants=("y_pred",))                    # :1044-1068
    run("attr", "get_attraction_chamfer", wants=("y_pred",))                     # :521-531
    run("emd", "get_emd", wants=("y_pred",))                                     # :990-1009
    # stroke_masks_metrics (metrics_handler.py:285-308 + utils/postprocessing.py:92-152) through the reference's own
    # MetricsHandler, in a fresh interpreter: the real `utils` package must not meet the stand-ins loss_handler_module() installed
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="mp_g13_")
    np.savez(os.path.join(tmp, "in.npz"), masks=masks, scores=scores)
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {ROOT!r})
from oracle import env_stubs, ref_import as R
env_stubs.install(); R.install_pytorch3d_stub(); R.neutralise_cuda()
sys.path.insert(0, {R.REF_ROOT!r})
from metrics_handler import MetricsHandler
d = np.load({os.path.join(tmp, 'in.npz')!r})
mh = MetricsHandler(config=dict(extra_data=['orientnorm'], lambda_points=4), metrics=['stroke_masks_metrics'])
out = mh.compute(n_strokes=[6, 5, 6], pred_stroke_masks=torch.from_numpy(d['masks']), mask_scores=torch.from_numpy(d['scores']))
np.save({os.path.join(tmp, 'out.npy')!r}, np.asarray(out, dtype=np.float64))
"""
    subprocess.run([sys.executable, "-c", code], check=True, cwd=R.REF_ROOT)
    metric_values = np.load(os.path.join(tmp, "out.npy"))
    cases.update(metric_n_strokes=np.array([6, 5, 6]), metric_values=metric_values)
    save("g13_losses", **cases)


def g14_collate():
    """The reference collate function (utils/dataset/paintnet_ODv1.py:713-847) on ragged synthetic samples: the padded
    MaskPlanner tensors the training loop consumes."""
    print("g14_collate")
    from oracle import env_stubs
    env_stubs.install()
    # g7 / g11 / g13 load loss_handler.py against STAND-INS for the reference's `utils` / `models` packages (ref_import.
    # loss_handler_module); this fixture needs the real `utils` package, so the stand-ins step aside while it is imported and come
    # back afterwards (the all-in-one run `python -m oracle.gen_golden` otherwise died here with ModuleNotFoundError: utils.dataset)
    parked = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "utils" or k.startswith("utils.")}
    sys.path.insert(0, R.REF_ROOT)
    try:
        from utils.dataset.paintnet_ODv1 import Paintnet_ODv1_CollateBatch
    finally:
        sys.path.remove(R.REF_ROOT)
        for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
            del sys.modules[k]
        sys.modules.update(parked)
    rng = np.random.default_rng(1414)
    cfg = R.AttrDict(lambda_points=4, overlapping=1, extra_data=["orientnorm"], task_name="MaskPlanner", out_prototypes=None,
                     load_extra_data=["stroke_masks"], traj_with_equally_spaced_points=True)      # traj_sampling_v2.yaml
    collate = Paintnet_ODv1_CollateBatch(cfg)
    samples, cases = [], {}
    for i, (n_pts, n_seg, n_str) in enumerate([(40, 12, 3), (71, 21, 5), (1, 1, 1), (64, 20, 4)]):
        smp = dict(point_cloud=rng.normal(size=(256, 3)).astype(np.float32),
                   traj=rng.normal(size=(n_seg, 24)).astype(np.float32),
                   traj_as_pc=rng.normal(size=(n_pts, 6)).astype(np.float32),
                   stroke_ids=np.sort(rng.integers(0, n_str, size=n_seg)).astype(np.float32),
                   stroke_ids_as_pc=np.sort(rng.integers(0, n_str, size=n_pts)).astype(np.float32),
                   stroke_masks=(rng.uniform(size=(n_str, n_seg)) > 0.5).astype(np.float32),
                   dirname=f"sample_{i}", n_strokes=n_str)
        samples.append(smp)
        for k, v in smp.items():
            if isinstance(v, np.ndarray):
                cases[f"in{i}_{k}"] = v
    out = collate([dict(s) for s in samples])
    for k, v in out.items():
        if torch.is_tensor(v):
            cases["out_" + k] = v.numpy()
        elif k == "stroke_masks":
            for i, m in enumerate(v):
                cases[f"out_stroke_masks{i}"] = m.numpy()
    cases["none_keys"] = np.array(sorted(k for k, v in out.items() if v is None))
    cases["n_strokes"] = np.array(out["n_strokes"])
    cases["n_samples"] = np.int64(len(samples))
    save("g14_collate", **cases)


def g15_train():
    print("g15_train")
    pc = R.pointnet2_cls_ssg()
    pu = R.pointnet2_utils()
    g5 = np.load(os.path.join(OUT, "g5_model.npz"))
    model = pc.PointNet2Regressor_StrokeMasks(outdim=12, outdim_orient=12, weight_orient=0.25, out_vectors=99,
                                              hidden_size=(64, 64), pred_stroke_masks=True, n_stroke_masks=6,
                                              mask_confidence_scores=True, segment_confidence_scores=False)
    model.load_state_dict({k[3:]: torch.from_numpy(g5[k].copy()) for k in g5.files if k.startswith("sd_")}, strict=True)
    model.train()
    model.dropout.p = 0.0                      # train-mode BatchNorm everywhere, no dropout draw
    rng = np.random.default_rng(1515)
    B = 8
    xyz = syn.point_cloud(rng, B, 1024, "cuboid")
    s1, _ = ref_fps(pu, xyz, 512, 77)
    torch.manual_seed(77)
    _ = torch.randint(0, 1024, (B,))
    s2 = torch.randint(0, 512, (B,)).numpy()
    w_out = rng.normal(size=(B, 99, 24)).astype(np.float32)
    w_sm = rng.normal(size=(B, 6, 99)).astype(np.float32)
    torch.manual_seed(77)
    out, sm_out, mask_conf, _ = model(torch.from_numpy(xyz).permute(0, 2, 1))
    ((out * torch.from_numpy(w_out)).sum() + (sm_out * torch.from_numpy(w_sm)).sum() + mask_conf.sum()).backward()
    grads = {"grad_" + n: p.grad.numpy().copy() for n, p in model.named_parameters()
             if n in ("sa1.mlp_convs.1.weight", "sa1.mlp_bns.2.weight", "sa2.mlp_convs.2.weight", "sa2.mlp_bns.0.bias", "sa3.mlp_convs.0.weight",
                      "sa3.mlp_bns.1.weight", "fc1.weight", "bn2.weight", "fc3.weight", "fc_normals.bias", "sm_fc3.bias", "mask_conf_out.weight")}
    after = {"after_" + k: v.numpy().copy() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    save("g15_train", xyz=xyz, fps_start1=s1, fps_start2=s2, w_out=w_out, w_sm=w_sm, out=out.detach().numpy(),
         sm_out=sm_out.detach().numpy(), mask_conf=mask_conf.detach().numpy(), **grads, **after)


def g16_lambda():
    """The reference's segment builder on ragged strokes (one dropped for being shorter than lambda), overlapping 1 (the
    maskplanner configs) and 0 (centred windows), in a fresh interpreter with the real `utils` package."""
    print("g16_lambda")
    import subprocess
    import tempfile
    rng = np.random.default_rng(1616)
    cases = {}
    samples = []
    for i, lens in enumerate([(9, 4, 13), (3, 17, 6, 5), (25,), (7, 7, 2, 11, 4)]):
        n = sum(lens)
        poses = rng.normal(size=(n, 6)).astype(np.float32)
        ids = np.concatenate([np.full(L, s, dtype=np.float32) for s, L in enumerate(lens)])
        samples.append((poses, ids))
        cases[f"poses{i}"], cases[f"ids{i}"] = poses, ids
    tmp = tempfile.mkdtemp(prefix="mp_g16_")
    np.savez(os.path.join(tmp, "in.npz"), **cases)
    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
from oracle import env_stubs
env_stubs.install()
sys.path.insert(0, {R.REF_ROOT!r})
from utils.pointcloud import get_sequences_of_lambda_points
d = np.load({os.path.join(tmp, 'in.npz')!r})
out = {{}}
for i in range({len(samples)}):
    for lam, ov in ((4, 1), (4, 0), (3, 2)):
        t, s = get_sequences_of_lambda_points(d[f'poses{{i}}'].copy(), d[f'ids{{i}}'].copy(), lam, 'x', overlapping=ov, extra_data=['orientnorm'])
        out[f'traj{{i}}_{{lam}}_{{ov}}'], out[f'sid{{i}}_{{lam}}_{{ov}}'] = t.astype(np.float32), s.astype(np.float32)
np.savez({os.path.join(tmp, 'out.npz')!r}, **out)
"""
    subprocess.run([sys.executable, "-c", code], check=True, cwd=R.REF_ROOT)
    cases.update(dict(np.load(os.path.join(tmp, "out.npz"))))
    cases["n_samples"] = np.int64(len(samples))
    save("g16_lambda", **cases)


def _grad_digest(cases, prefix, named_params):
    """Gradients of a whole module in a small fixture: tensors up to 16 384 elements in full, larger ones as their L2 norm plus the
    values at 4 096 fixed positions (a seeded draw per tensor, repeated by the test)."""
    for i, (n, p) in enumerate(named_params):
        if p.grad is None:
            continue
        g = p.grad.detach().reshape(-1)
        if g.numel() <= 16384:
            cases[f"{prefix}grad_{n}"] = p.grad.numpy().copy()
        else:
            pos = np.random.default_rng(4242 + i).choice(g.numel(), size=4096, replace=False)
            cases[f"{prefix}gsamp_{n}"] = g.numpy()[pos].copy()
            cases[f"{prefix}gnorm_{n}"] = np.float64(g.double().norm())


def g17_siblings():
    """What round 2 left without a number: the three sibling regressors of models/pointnet2_cls_ssg.py (:85 SoPs, :177 3Dbbox, :463
    StrokeWise) in eval AND train mode with gradients, the BACKWARD of the two segmenters (models/pointnet2_seg.py:14-96, 258-339;
    g9 holds their eval forward), and sample_and_group(returnfps=True) (models/pointnet2_utils.py:144-145)."""
    print("g17_siblings")
    pc = R.pointnet2_cls_ssg()
    pu = R.pointnet2_utils()
    sg = R.pointnet2_seg()
    rng = np.random.default_rng(1717)
    cases = {}
    B, N = 4, 1024
    xyz = syn.point_cloud(rng, B, N, "cuboid")
    cases["xyz"] = xyz

    def draws(seed, n1=N, n2=512):
        torch.manual_seed(seed)
        return torch.randint(0, n1, (B,)).numpy(), torch.randint(0, n2, (B,)).numpy()

    models = {
        "sops": (lambda: pc.PointNet2Regressor_SoPs(out_vectors=7, outdim=3, outdim_orient=3, weight_orient=0.25, hidden_size=(64, 64),
                                                    sop_confidence_scores=True), 171),
        "bbox": (lambda: pc.PointNet2Regressor_3Dbbox(out_bboxes=5, hidden_size=(64, 64)), 172),
        "sw": (lambda: pc.PointNet2Regressor_StrokeWise(outdim=12, outdim_orient=12, weight_orient=0.25, out_vectors=9, hidden_size=(64, 64),
                                                        stroke_confidence_scores=True, point_confidence_scores=True,
                                                        n_points_per_out_vector=4), 173),
    }
    for tag, (ctor, seed) in models.items():
        torch.manual_seed(seed)
        m = ctor()
        cases.update({f"{tag}_ck_{k}": np.float64(v.double().abs().sum()) for k, v in m.state_dict().items()})
        x = torch.from_numpy(xyz).permute(0, 2, 1)
        m.eval()
        cases[f"{tag}_eval_s1"], cases[f"{tag}_eval_s2"] = draws(seed + 100)
        torch.manual_seed(seed + 100)
        with torch.no_grad():
            outs = m(x)
        outs = outs if isinstance(outs, tuple) else (outs,)
        for i, o in enumerate(outs):
            cases[f"{tag}_eval_out{i}"] = o.numpy()
        m.train()
        m.dropout.p = 0.0
        cases[f"{tag}_train_s1"], cases[f"{tag}_train_s2"] = draws(seed + 200)
        torch.manual_seed(seed + 200)
        outs = m(x)
        outs = outs if isinstance(outs, tuple) else (outs,)
        total = 0
        for i, o in enumerate(outs):
            w = rng.normal(size=tuple(o.shape)).astype(np.float32)
            cases[f"{tag}_train_out{i}"], cases[f"{tag}_w{i}"] = o.detach().numpy().copy(), w
            total = total + (o * torch.from_numpy(w)).sum()
        total.backward()
        _grad_digest(cases, tag + "_", list(m.named_parameters()))
        cases.update({f"{tag}_after_{k}": v.detach().numpy().copy() for k, v in m.state_dict().items() if "running" in k})

    # segmenter backward: the modules of g9 (same seeds => same weights), train mode, a linear functional of the per-point output
    segs = rng.uniform(-0.5, 0.5, size=(B, 24, N)).astype(np.float32)
    cases["sg_in"] = segs
    for tag, ctor, seed, inp in (("pn", lambda: sg.PointNet2Segmenter_PaintNet_v1(inputdim=3, outdim_trasl=3, outdim_orient=3,
                                                                                  weight_orient=0.25, lambda_points=2), 9, xyz.transpose(0, 2, 1)),
                                 ("sg", lambda: sg.PointNet2Segmenter_v1(outdim=5, input_orient_dim=3, lambda_points=4,
                                                                         ball_in_xyz_space=True), 10, segs)):
        torch.manual_seed(seed)
        m = ctor()
        m.train()
        cases.update({f"{tag}_ck_{k}": np.float64(v.double().abs().sum()) for k, v in m.state_dict().items()})
        cases[f"{tag}_s1"], cases[f"{tag}_s2"] = draws(seed + 300)
        torch.manual_seed(seed + 300)
        out = m(torch.from_numpy(np.ascontiguousarray(inp)))
        w = rng.normal(size=tuple(out.shape)).astype(np.float32)
        cases[f"{tag}_out"], cases[f"{tag}_w"] = out.detach().numpy().copy(), w
        (out * torch.from_numpy(w)).sum().backward()
        _grad_digest(cases, tag + "_", list(m.named_parameters()))

    # sample_and_group(..., returnfps=True): the two extra return values
    feats = rng.normal(size=(B, N, 5)).astype(np.float32)
    cases["rf_feats"] = feats
    torch.manual_seed(77)
    cases["rf_start"] = torch.randint(0, N, (B,)).numpy()
    torch.manual_seed(77)
    new_xyz, new_points, grouped_xyz, fps_idx = pu.sample_and_group(64, 0.3, 16, torch.from_numpy(xyz), torch.from_numpy(feats), returnfps=True)
    cases.update(rf_new_xyz=new_xyz.numpy(), rf_new_points=new_points.numpy(), rf_grouped_xyz=grouped_xyz.numpy(), rf_fps_idx=fps_idx.numpy())
    save("g17_siblings", **cases)


def main():
    if not R.available():
        raise SystemExit("reference checkout not found; fixtures can only be generated in the build container")
    torch.set_num_threads(8)
    pu = R.pointnet2_utils()
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17"]
    if "g1" in which: g1_fps(pu)
    if "g2" in which: g2_bq(pu)
    if "g3" in which: g3_sa(pu)
    if "g4" in which: g4_msg(pu)
    if "g5" in which: g5_model()
    if "g6" in which: g6_cham()
    if "g7" in which: g7_mask()
    if "g8" in which: g8_hung()
    if "g9" in which: g9_seg()
    if "g10" in which: g10_fp(pu)
    if "g11" in which: g11_smooth()
    if "g12" in which: g12_flags()
    if "g13" in which: g13_losses()
    if "g14" in which: g14_collate()
    if "g15" in which: g15_train()
    if "g16" in which: g16_lambda()
    if "g17" in which: g17_siblings()


if __name__ == "__main__":
    main()
