"""Routing-conditioned gradient parity.

Every gradient check of the train-mode step used to be bounded at 1e-2 .. 4e-2 "because a gradient through a max-pool is discontinuous
in the forward values": two evaluations whose pre-pool activations differ by fp32 rounding pick another member of the few groups whose
two largest members are that close, and the gradient of such a group takes another route.  Plausible, but an explanation.  Here it is a
measurement: the CPU oracle (oracle/torch_ref.py, fp32 AND fp64) is evaluated with the discrete decisions TAKEN FROM THE HIP PATH -- the
max-pool arg-max every level stores for its backward pass (sa_mlp.ROUTE_TAP) and the nearest neighbours of the three chamfer terms
(ops.KNN_TAP; an arg-min is as discontinuous as an arg-max) -- so what remains between the two gradients is arithmetic -- and

  * every parameter gradient of the HIP path must agree with the fp64 decision-conditioned evaluation at fp32-rounding level
    (<= 4e-4 relative L2 for the encoder, <= 8e-4 anywhere; measured 0.5e-4 .. 4.5e-4) and be at most twice as far from it as the
    fp32 oracle evaluated with the same decisions,
  * the share of decisions the UNCONDITIONED fp32 oracle takes differently is counted (3e-4 of the pool routes, 1e-4 of the
    neighbours, ~1e-6 of the ReLU masks): sqrt(share) is the size of the unconditioned distance, and sharing them removes it.
Three kinds of decision matter, and the first round of this test found them one by one: the pool arg-max alone left 3e-3, the
neighbours of the chamfer terms took the heads to 1e-4, the ReLU masks (an activation within rounding of 0) took the encoder there.
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle(model_sd, batch, cat, cfg, dtype, routes=None, argmax_out=None, nn_routes=None, nn_out=None, relu_masks=None, head_masks=None):
    from oracle import torch_ref as T
    sd = {}
    for k, v in model_sd.items():
        t = v.detach().clone()
        if t.dtype.is_floating_point:
            t = t.to(dtype)
            if "running" not in k:
                t.requires_grad_(True)
        sd[k] = t
    starts = [s.numpy() for s in batch["fps_start"]]
    out, sm, conf, feat = T.strokemasks_forward(sd, batch["point_cloud"].to(dtype), starts, train=True, out_vectors=cat.out_vectors,
                                                n_masks=cat.max_n_strokes, return_feat=True, routes=routes, argmax_out=argmax_out,
                                                relu_masks=relu_masks, head_masks=head_masks)
    loss = T.asymm_v6_loss(out, batch["traj"].to(dtype), sm, conf, batch["stroke_ids"], batch["traj_as_pc"].to(dtype), cfg,
                           nn_routes=nn_routes, nn_out=nn_out)
    loss.backward()
    return dict(feat=feat.detach(), loss=loss.detach(), grads={k: v.grad.detach() for k, v in sd.items() if v.requires_grad and v.grad is not None})


def _hip(model, batch, cfg):
    from maskplanner_amd import ops, sa_mlp
    from maskplanner_amd import pointnet2_utils as pu
    from maskplanner_amd.loss_handler import LossHandler
    lh = LossHandler(["asymm_v6_chamfer_with_stroke_masks"], cfg)
    det, ops.DETERMINISTIC = ops.DETERMINISTIC, True
    sa_mlp.ROUTE_TAP, ops.KNN_TAP, ops.RELU_TAP = [], [], []
    try:
        with pu.fps_start_override(batch["fps_start"]):
            feat = model.encode(batch["point_cloud"].cuda().permute(0, 2, 1))
            out, sm, conf, _ = model.heads(feat)
        routes = [r.clone() for r, _ in sa_mlp.ROUTE_TAP]
        # the ReLU masks of the nine encoder layers as the kernels form them: y = z * scale + shift (two roundings), y > 0
        masks = [[((z * sc + sh) > 0).cpu() for z, sc, sh in layers] for _, layers in sa_mlp.ROUTE_TAP]
        head_masks = [m.cpu() for m in ops.RELU_TAP]
        loss = lh.compute(return_list=False, y_pred=out, y=batch["traj"].cuda(), pred_stroke_masks=sm, mask_scores=conf,
                          seg_logits=None, stroke_ids=batch["stroke_ids"], traj_as_pc=batch["traj_as_pc"])
        nn = [i.clone().cpu() for i in ops.KNN_TAP]
        loss.backward()
    finally:
        ops.DETERMINISTIC = det
        sa_mlp.ROUTE_TAP, ops.KNN_TAP, ops.RELU_TAP = None, None, None
    grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    return dict(feat=feat.detach().cpu(), loss=loss.detach().cpu(), grads=grads), routes, nn, masks, head_masks


def _zero_exact(name):     # parameters whose exact gradient is 0 (tests/test_gpu_arbiter.py: _noise_only)
    return name.endswith("bias") and ("mlp_convs" in name or name in ("fc1.bias", "fc2.bias", "sm_fc1.bias", "sm_fc2.bias", "sa3.mlp_bns.2.bias"))


@pytest.mark.parametrize("B,N,hidden,seed", [(32, 5120, (256, 256), 17), (8, 1024, (64, 64), 1515)])
def test_gradients_agree_with_fp64_once_the_discrete_decisions_are_shared(oracle, monkeypatch, B, N, hidden, seed):
    """BASELINE configs[1] (cuboids, N = 5120, B = 32) and the 8-cloud shape of fixture g15, train mode, dropout off.
    [r4] measured (relative L2 of the parameter gradients, worst tensor): see profiles/r04_routing_*.json."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from maskplanner_amd import pointnet2_cls_ssg as pc
    from maskplanner_amd import synthetic as syn
    from maskplanner_amd.loss_handler import maskplanner_loss_config
    cat = syn.CATEGORIES["cuboids"]
    batch = syn.make_batch(seed, B, N, "cuboids", "cuboid")
    torch.manual_seed(4)
    model = pc.maskplanner_model(cat, hidden_size=hidden)
    model.dropout.p = 0.0
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = maskplanner_loss_config()
    monkeypatch.setenv("MP_RECOMPUTE_FIRST", "0")          # (every layer's raw activation stored: the masks are read off them)
    hip, routes, nn, masks, head_masks = _hip(model.cuda().train(), batch, cfg)
    assert len(routes) == 3 and len(nn) == 3 and len(head_masks) == 4 and all(len(m) == 3 for m in masks)
    S, K = (512, 128, 1), (32, 64, 128)
    routes = [r.cpu().to(torch.int64).view(B, s, -1) for r, s in zip(routes, S)]
    masks = [[m.view(B, s, k, -1) for m in lv] for lv, s, k in zip(masks, S, K)]
    own32, own_nn = [], []
    free32 = _oracle(sd0, batch, cat, cfg, torch.float32, argmax_out=own32, nn_out=own_nn)          # the oracle's own routing
    cond32 = _oracle(sd0, batch, cat, cfg, torch.float32, routes=routes, nn_routes=nn, relu_masks=masks, head_masks=head_masks)
    cond64 = _oracle(sd0, batch, cat, cfg, torch.float64, routes=routes, nn_routes=nn, relu_masks=masks, head_masks=head_masks)
    # 1. the routing is a legitimate arg-max: evaluated at the HIP path's members, the fp64 feature / loss equal the HIP values
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    assert rel(hip["feat"], cond64["feat"]) <= 1e-5 and rel(hip["loss"], cond64["loss"]) <= 1e-5
    # 2. how many (group, channel) pairs does the unconditioned fp32 oracle route differently?  (pairs whose pooled value is 0 -- every
    #    member clipped by the ReLU -- have no route at all and are not counted)
    share = []
    for mine, theirs in zip(own32, routes):
        share.append(float((mine != theirs).float().mean()))
    # ... and how many nearest neighbours of the three chamfer terms does it choose differently?  (rows beyond a cloud's length are 0 in both)
    nn_share = [float((a.to(torch.int64) != b.to(torch.int64)).float().mean()) for a, b in zip(own_nn, nn)]
    report = {"rerouted_share": share, "renearest_share": nn_share, "grads": {}}
    worst_c64, worst_c32, worst_free, behind = 0.0, 0.0, 0.0, []
    for n, g64 in cond64["grads"].items():
        if _zero_exact(n):
            continue
        e64, e32, efree = rel(hip["grads"][n], g64), rel(hip["grads"][n], cond32["grads"][n]), rel(hip["grads"][n], free32["grads"][n])
        eo = rel(cond32["grads"][n], g64)        # the fp32 ORACLE's own distance from the exact gradient, same decisions
        report["grads"][n] = dict(hip_vs_f64_conditioned=e64, oracle32_vs_f64_conditioned=eo, hip_vs_f32_conditioned=e32, hip_vs_f32_unconditioned=efree)
        worst_c64, worst_c32, worst_free = max(worst_c64, e64), max(worst_c32, e32), max(worst_free, efree)
        if e64 > 2.0 * eo + 2e-5:
            behind.append((n, e64, eo))
    report.update(worst_hip_vs_f64_conditioned=worst_c64, worst_hip_vs_f32_conditioned=worst_c32, worst_hip_vs_f32_unconditioned=worst_free)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"routing_B{B}_N{N}.json"), "w") as f:
            json.dump(report, f, indent=1)
    print("routing-conditioned:", {k: v for k, v in report.items() if k != "grads"})
    # 3. with the decisions shared, EVERY parameter gradient of the HIP path is at fp32-rounding distance from the fp64 evaluation
    #    ([r4] measured, B = 32: 0.5e-4 .. 2e-4 for every encoder and pose-head tensor, 4.5e-4 for fc_normals, 1e-5 for the mask head;
    #    the unconditioned comparison of the same step: 0.4e-2 .. 1.8e-2) ...
    assert worst_c64 <= 8e-4, sorted(((v["hip_vs_f64_conditioned"], k) for k, v in report["grads"].items()), reverse=True)[:6]
    enc = [v["hip_vs_f64_conditioned"] for k, v in report["grads"].items() if k.startswith("sa")]
    assert max(enc) <= 4e-4, max(enc)
    #    ... and at most twice as far from it as the fp32 oracle is (the arbiter's criterion, now free of tie-flipping luck: factor 2)
    assert not behind, behind[:6]
    # 4. the 1e-2 of the unconditioned comparison IS the discrete decisions: a few re-routed pool members / neighbours in 10^4, and
    #    sharing the decisions removes more than an order of magnitude
    assert max(share) <= 5e-3 and max(nn_share) <= 5e-2, (share, nn_share)
    assert worst_free >= 10.0 * worst_c64, (worst_free, worst_c64)
