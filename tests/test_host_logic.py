"""Host-side logic that needs no GPU: synthetic collated-batch contract, config plumbing, checkpoint-key
compatibility, drop-in aliasing, argument validation of the reference-shaped wrappers."""
import sys

import numpy as np
import pytest
import torch


def test_synthetic_batch_contract():
    """utils/dataset/paintnet_ODv1.py:824-845: everything f32, -100 / -1 padding, segment layout lambda*6."""
    from maskplanner_amd import synthetic as syn
    b = syn.make_batch(3, 4, 1024, "cuboids", "cuboid")
    assert b["point_cloud"].shape == (4, 1024, 3) and b["point_cloud"].dtype == torch.float32
    assert b["traj"].shape[2] == 24 and b["traj_as_pc"].shape[2] == 6 and b["stroke_ids"].dtype == torch.float32
    for i in range(4):
        ns, npnt = int(b["n_segments"][i]), int(b["n_points"][i])
        assert (b["traj"][i, ns:] == -100).all() and (b["traj"][i, :ns] != -100).any(dim=1).all()
        assert (b["stroke_ids"][i, ns:] == -1).all() and (b["stroke_ids"][i, :ns] >= 0).all()
        assert (b["traj_as_pc"][i, npnt:] == -100).all()
        # a segment is lambda=4 consecutive poses of one stroke, stride lambda-overlap=3
        seg0 = b["traj"][i, 0].view(4, 6)
        assert torch.equal(seg0, b["traj_as_pc"][i, :4])
        assert torch.equal(b["traj"][i, 1].view(4, 6)[0], b["traj_as_pc"][i, 3])
    assert b["traj"].shape[1] == int(b["n_segments"].max()) and b["traj_as_pc"].shape[1] == int(b["n_points"].max())
    for cat, c in syn.CATEGORIES.items():
        bb = syn.make_batch(1, 2, 256, cat, "ucube")
        assert 0 < int(bb["n_points"].min()) and int(bb["n_points"].max()) <= c.points_hi
    with pytest.raises(ValueError):
        syn.point_cloud(np.random.default_rng(0), 1, 8, "sphere")


def test_pose_dims_and_config_adapter():
    from maskplanner_amd import loss_handler as lh
    assert lh.get_dim_traj_points([]) == 3 and lh.get_dim_traj_points(["orientnorm"]) == 6
    assert lh.get_dim_traj_points(["orientquat"]) == 7
    with pytest.raises(ValueError):
        lh.get_dim_traj_points(["vel", "orientnorm"])
    cfg = lh._Config({"a": 1, "nested": 2})
    assert cfg["a"] == 1 and cfg.a == 1 and cfg.get("zzz", 5) == 5 and "a" in cfg.keys()

    class AttrCfg(dict):
        def __getattr__(self, k):
            return self[k]
    assert lh._Config(AttrCfg(x=3)).x == 3


def test_loss_handler_rejects_terms_outside_the_hot_path():
    from maskplanner_amd.loss_handler import LossHandler, maskplanner_loss_config
    cfg = maskplanner_loss_config()
    LossHandler(["asymm_v6_chamfer_with_stroke_masks"], cfg)
    with pytest.raises(NotImplementedError):
        LossHandler(["repulsion"], cfg)
    with pytest.raises(AssertionError):  # weight_<name> must exist, as in the reference (loss_handler.py:179-181)
        LossHandler(["emd"], cfg)


def test_state_dict_keys_match_the_reference(golden):
    """Checkpoint compatibility (test_maskplanner.py:162-188): same keys, shapes and registration order."""
    from maskplanner_amd import pointnet2_cls_ssg as pc
    g = golden("g5_model")
    m = pc.PointNet2Regressor_StrokeMasks(outdim=12, outdim_orient=12, weight_orient=0.25, out_vectors=99, hidden_size=(64, 64),
                                          pred_stroke_masks=True, n_stroke_masks=6, mask_confidence_scores=True)
    sd = m.state_dict()
    ref = [k[3:] for k in g.files if k.startswith("sd_")]
    assert list(sd.keys()) == ref  # np.savez preserves insertion order == the reference's registration order
    for k in ref:
        assert tuple(sd[k].shape) == tuple(g["sd_" + k].shape), k
    from maskplanner_amd import synthetic
    full = pc.maskplanner_model(synthetic.CATEGORIES["cuboids"])
    n = sum(p.numel() for p in full.parameters())
    assert abs(n - 35.74e6) < 0.02e6  # 35.74 M parameters for cuboids


def test_dropin_aliases_reference_module_names():
    from maskplanner_amd import dropin
    saved = {k: sys.modules.get(k) for k in list(dropin._ALIASES) + ["pytorch3d", "pytorch3d.ops"]}
    try:
        done = dropin.install()
        assert set(done) == set(dropin._ALIASES)
        import maskplanner_amd.pointnet2_utils as pu
        assert sys.modules["models.pointnet2_utils"] is pu
        from pytorch3d.ops.knn import knn_gather, knn_points  # the import line of pytorch3d_chamfer.py:12
        from pytorch3d.structures.pointclouds import Pointclouds  # noqa: F401  (:13)
        assert callable(knn_points) and callable(knn_gather)
        # every module-level def / class of models/pointnet2_utils.py (:9-279); models/pointnet2_seg.py:12 imports
        # PointNetFeaturePropagation by name, so a missing one breaks `import models` under the alias
        for name in ("timeit", "pc_normalize", "square_distance", "index_points", "farthest_point_sample", "query_ball_point",
                     "sample_and_group", "sample_and_group_all", "PointNetSetAbstraction", "PointNetSetAbstractionMsg",
                     "PointNetFeaturePropagation"):
            assert hasattr(pu, name), name
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_wrappers_validate_like_the_reference_before_touching_the_device():
    from maskplanner_amd.pytorch3d_chamfer import chamfer_distance
    x = torch.rand(2, 5, 3)
    with pytest.raises(ValueError, match="batch_reduction"):
        chamfer_distance(x, x, batch_reduction="max")
    with pytest.raises(ValueError, match="Expected points"):
        chamfer_distance(x[0], x)
    with pytest.raises(ValueError, match="should be either"):
        chamfer_distance([1, 2], x)
    from maskplanner_amd import knn
    with pytest.raises(ValueError, match="batch dimension"):
        knn.knn_points(torch.rand(2, 3, 3), torch.rand(3, 3, 3))


def test_set_abstraction_modules_have_reference_parameters():
    from maskplanner_amd import pointnet2_utils as pu
    sa = pu.PointNetSetAbstraction(512, 0.2, 32, 3, [64, 64, 128], False)
    assert [tuple(c.weight.shape) for c in sa.mlp_convs] == [(64, 3, 1, 1), (64, 64, 1, 1), (128, 64, 1, 1)]
    assert list(sa.state_dict())[:2] == ["mlp_convs.0.weight", "mlp_convs.0.bias"]
    msg = pu.PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [16, 32, 128], 0, [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
    assert "conv_blocks.2.1.weight" in msg.state_dict() and "bn_blocks.0.0.running_var" in msg.state_dict()
    assert tuple(msg.conv_blocks[0][0].weight.shape) == (32, 3, 1, 1)
    fp = pu.PointNetFeaturePropagation(384, [256, 256])
    assert tuple(fp.mlp_convs[0].weight.shape) == (256, 384, 1) and "mlp_bns.1.num_batches_tracked" in fp.state_dict()


def test_metrics_handler_bookkeeping_and_collate_validation_without_a_gpu(capsys):
    """The host side of the optional callers (metrics_handler.py:24-196; paintnet_ODv1.py:713-724): metric registry, output names,
    pretty printing, pose dimensionality, the renormalisation switch, and the extras a MaskPlanner collate does not take."""
    from maskplanner_amd.metrics_handler import MetricsHandler, get_dim_traj_points
    assert [get_dim_traj_points(e) for e in ([], ["vel"], ["orientnorm"], ["orientrotvec"], ["orientquat"])] == [3, 6, 6, 6, 7]
    with pytest.raises(ValueError):
        get_dim_traj_points(["vel", "orientnorm"])
    cfg = {"extra_data": ["orientnorm"], "lambda_points": 4, "normalization": "per-dataset"}
    mh = MetricsHandler(cfg, metrics=["pcd", "stroke_chamfer"])
    assert mh.tot_num_of_metrics() == 2 and mh.num_of_metrics("pcd") == 1 and mh.metric_index["chamfer_original"] == 1
    mh.pprint(np.array([1.234567, 2.0]), prefix="val")
    out = capsys.readouterr().out
    assert "point-wise chamfer distance: 1.23457" in out and "stroke chamfer distance: 2.0" in out
    assert MetricsHandler(cfg).compute() == 0                                   # no metrics requested (:126-127)
    with pytest.raises(AssertionError):
        mh.get_eval_metric("no_such_metric")
    with pytest.raises(NotImplementedError):
        mh.get_eval_metric("sop_metrics")
    assert not mh.renormalize_output and MetricsHandler(cfg, renormalize_output_config={"active": True, "from": 1, "to": 2}).renormalize_output
    with pytest.raises(AssertionError):                                         # renormalisation needs per-dataset normalisation (:113)
        MetricsHandler({**cfg, "normalization": "none"}, renormalize_output_config={"active": True})
    from maskplanner_amd.collate import Paintnet_ODv1_CollateBatch
    assert Paintnet_ODv1_CollateBatch({"load_extra_data": ["stroke_masks"]}).load_extra_data == ["stroke_masks"]
    with pytest.raises(NotImplementedError, match="stroke_prototypes"):
        Paintnet_ODv1_CollateBatch({"load_extra_data": ["stroke_prototypes"]})


def test_widen_interior_pads_with_dead_channels_and_keeps_the_parameters():
    """sa_mlp._widen_interior: an interior width between 64 and 128 becomes 128 -- zero weight rows / bias, gamma 1, beta 0,
    running (mean 0, var 1), zero columns in the next layer's weight; first / last widths and the other layers stay as they are;
    the padding is autograd-visible, so the gradients come back in the parameters' own shapes.  (Pure tensor logic: runs without a GPU.)"""
    import torch
    from maskplanner_amd import sa_mlp
    torch.manual_seed(0)
    widths, cin = [64, 96, 128], 4
    params, leaves, last = [], [], cin
    for c in widths:
        w = torch.randn(c, last, requires_grad=True)
        bias, gamma, beta = (torch.randn(c, requires_grad=True) for _ in range(3))
        rm, rv = torch.randn(c), torch.rand(c) + 0.5
        params += [w, bias, gamma, beta, rm, rv]
        leaves += [w, bias, gamma, beta]
        last = c
    orig = list(params)
    back = sa_mlp._widen_interior(params, widths)
    assert [tuple(p.shape) for p in params[0:6]] == [tuple(p.shape) for p in orig[0:6]]          # the 64-wide layer: untouched
    w1, b1, g1, be1, rm1, rv1 = params[6:12]
    assert w1.shape == (128, 64) and torch.equal(w1[:96], orig[6]) and not w1[96:].any()
    assert torch.equal(b1[:96], orig[7]) and not b1[96:].any()
    assert torch.equal(g1[:96], orig[8]) and torch.equal(g1[96:], torch.ones(32))
    assert torch.equal(be1[:96], orig[9]) and not be1[96:].any()
    assert torch.equal(rm1[:96], orig[10]) and not rm1[96:].any() and torch.equal(rv1[96:], torch.ones(32))
    w2 = params[12]
    assert w2.shape == (128, 128) and torch.equal(w2[:, :96], orig[12]) and not w2[:, 96:].any()
    assert [tuple(p.shape) for p in params[13:18]] == [tuple(p.shape) for p in orig[13:18]]      # the last layer's rows: untouched
    assert len(back) == 2 and back[0][0] is orig[10] and back[1][0] is orig[11]                  # running stats to write back
    (w1.sum() + g1.sum() * 2.0 + w2.sum() * 3.0 + b1.sum() + be1.sum()).backward()
    for p_, want in ((orig[6], 1.0), (orig[8], 2.0), (orig[12], 3.0), (orig[7], 1.0), (orig[9], 1.0)):
        assert p_.grad.shape == p_.shape and torch.equal(p_.grad, torch.full_like(p_, want))
    # nothing to widen: the list is returned as it was
    p2 = list(orig[:6]) + list(orig[12:18])
    same = list(p2)
    assert sa_mlp._widen_interior(p2, [64, 128]) == [] and all(a is b for a, b in zip(p2, same))


def test_fast_zero_grad_and_the_flattened_module_tree():
    """[r6] The drop-in models answer `zero_grad()` (twice per iteration of the reference's loop, train_maskplanner.py:183, 226) and this package's
    per-call look at the module tree from ONE flattened list, re-validated on every use: same effect as nn.Module.zero_grad / parameters() /
    modules() after parameters and submodules were replaced through registration, through the owner's dict, or added later; a copy of the model
    starts without the cache; nothing of it reaches the state_dict."""
    import copy
    import torch
    from maskplanner_amd import graphed, synthetic
    from maskplanner_amd.pointnet2_cls_ssg import maskplanner_model
    m = maskplanner_model(synthetic.CATEGORIES["cuboids"], hidden_size=(64, 64))

    def dirty(model):
        for p in model.parameters():
            p.grad = torch.ones_like(p)

    def clean(model):
        return all(p.grad is None for p in model.parameters())
    dirty(m); m.zero_grad(); assert clean(m)
    n_train, n_req, hooked, ptrs = graphed._walk(m)
    assert n_train == sum(1 for x in m.modules() if x.training) and n_req == sum(1 for p in m.parameters() if p.requires_grad)
    assert not hooked and ptrs == tuple(p.data_ptr() for _, _, p in graphed._flat(m).pars) and len(ptrs) == sum(1 for _ in m.parameters())
    m.fc3.weight = torch.nn.Parameter(torch.zeros_like(m.fc3.weight))                     # through registration (the global hook bumps the version)
    m.fc2._parameters["weight"] = torch.nn.Parameter(torch.zeros_like(m.fc2.weight))      # behind registration's back (caught per entry)
    m.extra = torch.nn.Linear(4, 4)                                                       # a new submodule
    dirty(m); m.zero_grad(); assert clean(m)
    assert graphed._walk(m)[1] == sum(1 for p in m.parameters() if p.requires_grad)
    m.sa2.eval(); m.fc1.weight.requires_grad_(False)
    n_train, n_req, _, _ = graphed._walk(m)
    assert n_train == sum(1 for x in m.modules() if x.training) and n_req == sum(1 for p in m.parameters() if p.requires_grad)
    h = m.sa1.register_forward_hook(lambda *a: None)
    assert graphed._walk(m)[2]
    h.remove()
    h = m.fc3.weight.register_hook(lambda g: g)
    assert graphed._walk(m)[2]
    h.remove()
    assert not graphed._walk(m)[2]
    m2 = copy.deepcopy(m)
    assert m2.__dict__["_mp_flat"].version == -1
    dirty(m2); m2.zero_grad(); assert clean(m2)
    dirty(m); m.zero_grad(set_to_none=False)
    assert all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in m.parameters())
    assert not any("_mp" in k for k in m.state_dict())
