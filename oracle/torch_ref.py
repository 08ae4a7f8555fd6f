"""fp32 CPU restatement of the floating-point half of the hot path -- TEST INFRASTRUCTURE ONLY.

Index-producing steps (FPS, ball query, kNN, LAP) come from the C oracle (oracle/mp_oracle.c);
the floating-point algebra around them (shared MLP + BatchNorm + max-pool, heads, chamfer
reductions, stroke-mask loss) is written here with plain torch CPU ops and autograd.  Checked
against the golden fixtures g3..g7 (tests/test_oracle_golden.py); used as the checker for the
HIP path at sizes the fixtures do not cover and as bench.py's cpu_baseline ("port").

Layout note: activations are kept positions-major [B,S,K,C] (the reference uses [B,C,K,S]); a 1x1
Conv2d is then a plain matmul over the last axis, and BatchNorm2d statistics are over all leading axes.

fp64 arbiter: every function takes its floating-point type from its inputs.  Called with float64 weights and
inputs, the algebra runs in double precision while every DISCRETE step (FPS, ball query, nearest neighbours,
the assignment) is still taken on the float32 values the fp32 paths see -- the same function with ~1e-16
rounding, against which the fp32 oracle and the HIP path can both be measured (tests/test_gpu_arbiter.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ------------------------------------------------------------------------------------------------
# set abstraction  (models/pointnet2_utils.py:112-216)
# ------------------------------------------------------------------------------------------------
def _bn_train(z, gamma, beta, running=None, stored=None):
    """BatchNorm (training mode) over all axes but the last; biased var for normalisation,
    unbiased var for the running estimate (torch.nn.BatchNorm2d semantics).
    stored: the value that is normalised when it differs from the one the statistics are taken from (bf16 activation storage)."""
    red = tuple(range(z.ndim - 1))
    mean = z.mean(red)
    var = z.var(red, unbiased=False)
    if running is not None:
        n = z.numel() // z.shape[-1]
        rm, rv = running
        with torch.no_grad():
            rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
            rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var * n / max(n - 1, 1))
    return ((z if stored is None else stored) - mean) / torch.sqrt(var + BN_EPS) * gamma + beta


def _bn_eval(z, gamma, beta, rm, rv):
    return (z - rm) / torch.sqrt(rv + BN_EPS) * gamma + beta


def _r16(t):
    """Round to bf16 (nearest even) and back: the value a bf16 matrix-core operand carries."""
    return t.to(torch.bfloat16).to(torch.float32)


class _Bf16Matmul(torch.autograd.Function):
    """a [..., Ci] x w [Co, Ci] -> [..., Co] as the bf16 grouped-MLP variant defines it (mp_sa_mlp_*_bf16): BOTH operands of
    every contraction rounded to bf16, products and sums in fp32 (a product of two bf16 values is exact in fp32) --
    forward a.w^T, backward g.w (input gradient) and g^T.a (weight gradient), each with its two operands rounded."""

    @staticmethod
    def forward(ctx, a, w):
        ctx.save_for_backward(a, w)
        return _r16(a) @ _r16(w).t()

    @staticmethod
    def backward(ctx, g):
        a, w = ctx.saved_tensors
        g16 = _r16(g)
        ga = g16 @ _r16(w)
        gw = g16.reshape(-1, g16.shape[-1]).t() @ _r16(a).reshape(-1, a.shape[-1])
        return ga, gw


class _StoreRound(torch.autograd.Function):
    """The bf16 variant's ACTIVATION STORAGE (csrc/sa_mlp.hip: chain_store16): an interior layer's raw activation lives in memory as
    bf16 -- the next layer (and the backward pass) see the rounded value; the gradient passes straight through."""

    @staticmethod
    def forward(ctx, z):
        return _r16(z)

    @staticmethod
    def backward(ctx, g):
        return g


class _GradRound(torch.autograd.Function):
    """... and the gradient with respect to a layer's input is stored as bf16 too."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _r16(g)


def bf16_storage(first_kind, widths, K):
    """oracle twin of mp_sa_mlp_bf16_storage: first_kind 1 = first layer fed by bare coordinates (recomputed by the kernels), 2 = fed by
    coordinates + features (factorised); widths = the layers' output widths (an interior width in (64, 128) is carried as 128)."""
    w = [128 if (64 < c < 128 and i < len(widths) - 1) else c for i, c in enumerate(widths)]
    if len(w) < 2 or K not in (32, 64, 128):
        return False
    if first_kind == 1 and not (len(w) >= 3 and w[0] == 64 and w[1] in (64, 128)):
        return False
    if first_kind == 2 and w[0] not in (64, 128, 256):
        return False
    for ci, co in zip(w[:-1], w[1:]):
        if ci not in (64, 128) or co not in (64, 128, 256) or (co == 64 and ci == 128) or (co == 256 and ci != 128):
            return False
    return first_kind in (1, 2)


def shared_mlp_max(x, layers, train, bf16=False, route=None, argmax_out=None, relu_masks=None, store16=False):
    """x [B,S,K,Cin] -> [B,S,Cout]: (1x1 conv, BN, ReLU) x len(layers), then max over K.
    layers: list of dict(weight[Co,Ci], bias[Co], gamma, beta, running_mean, running_var).
    bf16: the contraction of every layer with bf16-rounded operands (_Bf16Matmul); everything else fp32.
    route [B,S,Cout] (int64, member index inside the group): ROUTING-CONDITIONED evaluation -- the pooled value of a (group, channel)
    is read at the given member instead of at this evaluation's own arg-max, so that two implementations whose forward values
    differ by rounding backpropagate through the SAME members (a max-pool's gradient is discontinuous at ties: with the routing taken
    from the implementation under test, the gradients are comparable at rounding level).  argmax_out (list): receives this
    evaluation's own arg-max [B,S,Cout], for counting the (group, channel) pairs the two implementations route differently.
    relu_masks: per layer a bool tensor [B,S,K,Co] -- the ReLU of that layer as `z * mask` with the mask of the other implementation
    (an activation within rounding of 0 is as much a discrete decision as an arg-max).
    store16 (with bf16; 1: first layer recomputed, 2: first layer factorised -- bf16_storage()'s first_kind): the chain keeps its activations in memory as bf16 -- BatchNorm statistics from the unrounded z, the normalised
    value (and everything downstream) from the rounded one; the pooled output of the LAST layer comes from the unrounded accumulators;
    gradients with respect to the interior activations rounded likewise."""
    for li, L in enumerate(layers):
        if store16 and bf16 and li > 0:
            x = _GradRound.apply(x)
        z0 = _Bf16Matmul.apply(x, L["weight"]) if bf16 else x @ L["weight"].t()
        z = z0 + L["bias"]
        # what is stored is the BIAS-FREE product (the conv bias is folded into the BatchNorm shift);
        # store16 == 1: the first layer is RECOMPUTED by its consumers from the input rows -- never stored, never rounded
        zs = (_StoreRound.apply(z0) + L["bias"]) if (store16 and bf16 and li < len(layers) - 1 and not (li == 0 and store16 == 1)) else None
        if train:
            z = _bn_train(z, L["gamma"], L["beta"], (L["running_mean"], L["running_var"]), stored=zs)
        else:
            z = _bn_eval(z if zs is None else zs, L["gamma"], L["beta"], L["running_mean"], L["running_var"])
        x = torch.relu(z) if relu_masks is None else z * relu_masks[li].to(z.dtype)
    if argmax_out is not None:
        argmax_out.append(x.detach().max(dim=2)[1])
    if route is not None:
        return torch.gather(x, 2, route.to(torch.int64)[:, :, None, :]).squeeze(2)
    return x.max(dim=2)[0]


def layers_from_state(sd, prefix, convs="mlp_convs", bns="mlp_bns"):
    """Collect SA layer tensors from a reference-layout state_dict (SURVEY section 5 key list)."""
    out = []
    i = 0
    while f"{prefix}{convs}.{i}.weight" in sd:
        w = sd[f"{prefix}{convs}.{i}.weight"]
        out.append(dict(
            weight=w.reshape(w.shape[0], w.shape[1]), bias=sd[f"{prefix}{convs}.{i}.bias"],
            gamma=sd[f"{prefix}{bns}.{i}.weight"], beta=sd[f"{prefix}{bns}.{i}.bias"],
            running_mean=sd[f"{prefix}{bns}.{i}.running_mean"], running_var=sd[f"{prefix}{bns}.{i}.running_var"]))
        i += 1
    return out


def set_abstraction(xyz, feats, layers, npoint, radius, nsample, fps_start, train, group_all=False, bf16=False, route=None, argmax_out=None,
                    relu_masks=None):
    """xyz [B,N,3], feats [B,N,D] or None (points-major).  Returns new_xyz [B,S,3], new_feats [B,S,C'].
    route / argmax_out: see shared_mlp_max."""
    B, N, _ = xyz.shape
    if group_all:  # sample_and_group_all :151-168 -- xyz NOT centred, new_xyz = 0
        x = xyz if feats is None else torch.cat([xyz, feats], -1)
        return torch.zeros(B, 1, 3, dtype=xyz.dtype), shared_mlp_max(x[:, None], layers, train, bf16, route, argmax_out, relu_masks)
    xyz_np = xyz.detach().float().numpy()
    fidx = O.fps(xyz_np, npoint, fps_start)
    new_xyz_np = O.index_points(xyz_np, fidx)
    gidx = torch.from_numpy(O.ball_query(radius, nsample, xyz_np, new_xyz_np))
    new_xyz = torch.from_numpy(new_xyz_np).to(xyz.dtype)
    bidx = torch.arange(B)[:, None, None]
    g = xyz[bidx, gidx] - new_xyz[:, :, None]
    if feats is not None:
        g = torch.cat([g, feats[bidx, gidx]], -1)  # xyz channels first (:138)
    kind = 1 if feats is None else 2
    st16 = kind if (bf16 and bf16_storage(kind, [L["weight"].shape[0] for L in layers], nsample)) else 0
    return new_xyz, shared_mlp_max(g, layers, train, bf16, route, argmax_out, relu_masks, store16=st16)


def set_abstraction_msg(xyz, feats, blocks, npoint, radii, nsamples, fps_start, train, bf16=False):
    """PointNetSetAbstractionMsg (:219-276): one FPS, per-radius ball query; channel order FEATS first, xyz last."""
    B = xyz.shape[0]
    xyz_np = xyz.detach().float().numpy()
    fidx = O.fps(xyz_np, npoint, fps_start)
    new_xyz_np = O.index_points(xyz_np, fidx)
    new_xyz = torch.from_numpy(new_xyz_np).to(xyz.dtype)
    bidx = torch.arange(B)[:, None, None]
    outs = []
    for layers, r, K in zip(blocks, radii, nsamples):
        gidx = torch.from_numpy(O.ball_query(r, K, xyz_np, new_xyz_np))
        g = xyz[bidx, gidx] - new_xyz[:, :, None]
        if feats is not None:
            g = torch.cat([feats[bidx, gidx], g], -1)
        kind = 1 if feats is None else 2
        st16 = kind if (bf16 and bf16_storage(kind, [L["weight"].shape[0] for L in layers], K)) else 0
        outs.append(shared_mlp_max(g, layers, train, bf16, store16=st16))
    return new_xyz, torch.cat(outs, -1)


# ------------------------------------------------------------------------------------------------
# full model  (models/pointnet2_cls_ssg.py:233-344, eval or train without dropout)
# ------------------------------------------------------------------------------------------------
def _bn1d(x, sd, name, train):
    g, b = sd[name + ".weight"], sd[name + ".bias"]
    if train:
        return _bn_train(x, g, b, (sd[name + ".running_mean"], sd[name + ".running_var"]))
    return _bn_eval(x, g, b, sd[name + ".running_mean"], sd[name + ".running_var"])


def msg_blocks_from_state(sd, prefix):
    """The per-scale layer lists of a PointNetSetAbstractionMsg (state_dict keys conv_blocks.{i}.{j}.*, bn_blocks.{i}.{j}.*)."""
    out = []
    i = 0
    while f"{prefix}conv_blocks.{i}.0.weight" in sd:
        out.append(layers_from_state(sd, prefix, convs=f"conv_blocks.{i}", bns=f"bn_blocks.{i}"))
        i += 1
    return out


# the multi-scale encoder of pointnet2_cls_ssg.PointNet2Regressor_StrokeMasks_MSG (upstream pointnet2_cls_msg widths)
MSG_LEVELS = ((512, (0.1, 0.2, 0.4), (16, 32, 128)), (128, (0.2, 0.4, 0.8), (32, 64, 128)))


def encoder_forward(sd, xyz, fps_starts, train, encoder="ssg", bf16=False, routes=None, argmax_out=None, relu_masks=None):
    """xyz [B,N,3] -> global feature [B,1024]: sa1 -> sa2 -> sa3 of the SSG (models/pointnet2_cls_ssg.py:266-268, 299-307) or
    the multi-scale encoder; bf16: the grouped-MLP contractions with bf16-rounded operands.
    routes (SSG only): the three levels' max-pool routing [r1, r2, r3] (see shared_mlp_max); argmax_out: list receiving this
    evaluation's own three arg-max tensors."""
    B = xyz.shape[0]
    r = routes if routes is not None else (None, None, None)
    m = relu_masks if relu_masks is not None else (None, None, None)       # per level: the three layers' ReLU masks
    if encoder == "msg":
        (n1, r1, k1), (n2, r2, k2) = MSG_LEVELS
        l1_xyz, l1 = set_abstraction_msg(xyz, None, msg_blocks_from_state(sd, "sa1."), n1, r1, k1, fps_starts[0], train, bf16)
        l2_xyz, l2 = set_abstraction_msg(l1_xyz, l1, msg_blocks_from_state(sd, "sa2."), n2, r2, k2, fps_starts[1], train, bf16)
    else:
        l1_xyz, l1 = set_abstraction(xyz, None, layers_from_state(sd, "sa1."), 512, 0.2, 32, fps_starts[0], train, bf16=bf16, route=r[0], argmax_out=argmax_out, relu_masks=m[0])
        l2_xyz, l2 = set_abstraction(l1_xyz, l1, layers_from_state(sd, "sa2."), 128, 0.4, 64, fps_starts[1], train, bf16=bf16, route=r[1], argmax_out=argmax_out, relu_masks=m[1])
    _, l3 = set_abstraction(l2_xyz, l2, layers_from_state(sd, "sa3."), None, None, None, None, train, group_all=True, bf16=bf16,
                            route=r[2] if encoder != "msg" else None, argmax_out=argmax_out if encoder != "msg" else None,
                            relu_masks=m[2] if encoder != "msg" else None)
    return l3.reshape(B, -1)


def strokemasks_forward(sd, xyz, fps_starts, train, out_vectors, n_masks, weight_orient=0.25, dropout_masks=None,
                        encoder="ssg", bf16=False, return_feat=False, routes=None, argmax_out=None, relu_masks=None, head_masks=None):
    """sd: reference-layout state_dict of tensors; xyz [B,N,3].  dropout_masks: optional list of 4
    pre-scaled keep masks (train mode) so dropout is reproducible; None = no dropout.
    encoder "msg": two multi-scale levels + group_all; bf16: the grouped-MLP contractions with bf16-rounded operands."""
    B = xyz.shape[0]
    feat = encoder_forward(sd, xyz, fps_starts, train, encoder, bf16, routes, argmax_out, relu_masks)
    act = (lambda z, i: torch.relu(z)) if head_masks is None else (lambda z, i: z * head_masks[i].to(z.dtype))     # head_masks: bn1, bn2, sm_bn1, sm_bn2
    dm = dropout_masks or [1.0, 1.0, 1.0, 1.0]
    x = act(_bn1d(feat @ sd["fc1.weight"].t() + sd["fc1.bias"], sd, "bn1", train), 0) * dm[0]
    final = act(_bn1d(x @ sd["fc2.weight"].t() + sd["fc2.bias"], sd, "bn2", train), 1) * dm[1]
    pos = final @ sd["fc3.weight"].t() + sd["fc3.bias"]
    s1 = act(_bn1d(feat @ sd["sm_fc1.weight"].t() + sd["sm_fc1.bias"], sd, "sm_bn1", train), 2) * dm[2]
    s2 = act(_bn1d(s1 @ sd["sm_fc2.weight"].t() + sd["sm_fc2.bias"], sd, "sm_bn2", train), 3) * dm[3]
    sm_out = (s2 @ sd["sm_fc3.weight"].t() + sd["sm_fc3.bias"]).view(B, n_masks, -1)
    mask_conf = s2 @ sd["mask_conf_out.weight"].t() + sd["mask_conf_out.bias"]
    nrm = torch.tanh(final @ sd["fc_normals.weight"].t() + sd["fc_normals.bias"]).view(B, -1, 3)
    nrm = F.normalize(nrm, dim=-1) * weight_orient
    out = torch.cat([pos.view(B, -1, 3), nrm], -1).view(B, out_vectors, -1)
    if return_feat:
        return out, sm_out, mask_conf, feat
    return out, sm_out, mask_conf


def _pose_out(sd, final, pos, B, out_vectors, weight_orient):
    """The pose assembly shared by the regressors (e.g. models/pointnet2_cls_ssg.py:159-166): unit normals x weight_orient
    appended to each predicted position."""
    nrm = torch.tanh(final @ sd["fc_normals.weight"].t() + sd["fc_normals.bias"]).view(B, -1, 3)
    nrm = F.normalize(nrm, dim=-1) * weight_orient
    return torch.cat([pos.view(B, -1, 3), nrm], -1).view(B, out_vectors, -1)


def sibling_forward(kind, sd, xyz, fps_starts, train, out_vectors, weight_orient=0.25, n_points=None):
    """The three sibling regressors that share the StrokeMasks encoder and trunk (dropout off):
    "sops"  PointNet2Regressor_SoPs       (models/pointnet2_cls_ssg.py:85-174)  -> (out, sop_conf)
    "bbox"  PointNet2Regressor_3Dbbox     (:177-230)                            -> (out,)
    "sw"    PointNet2Regressor_StrokeWise (:463-559)                            -> (out, point_conf, stroke_conf)"""
    B = xyz.shape[0]
    feat = encoder_forward(sd, xyz, fps_starts, train)
    x = torch.relu(_bn1d(feat @ sd["fc1.weight"].t() + sd["fc1.bias"], sd, "bn1", train))
    last = torch.relu(_bn1d(x @ sd["fc2.weight"].t() + sd["fc2.bias"], sd, "bn2", train))
    pos = last @ sd["fc3.weight"].t() + sd["fc3.bias"]
    if kind == "bbox":
        return (pos.view(B, out_vectors, 6),)
    out = _pose_out(sd, last, pos, B, out_vectors, weight_orient)
    if kind == "sops":
        return out, last @ sd["sop_conf_out.weight"].t() + sd["sop_conf_out.bias"]
    stroke = last @ sd["stroke_conf_out.weight"].t() + sd["stroke_conf_out.bias"]
    point = (last @ sd["point_conf_out.weight"].t() + sd["point_conf_out.bias"]).view(B, out_vectors, n_points)
    return out, point, stroke


# ------------------------------------------------------------------------------------------------
# chamfer  (pytorch3d_chamfer.py:76-344: the flags the maskplanner path exercises)
# ------------------------------------------------------------------------------------------------
class _Knn1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p1, p2, l1, l2):
        d, i = O.knn_points(p1.detach().numpy(), p2.detach().numpy(), l1.numpy(), l2.numpy(), 1)
        d, i = torch.from_numpy(d[..., 0].copy()), torch.from_numpy(i[..., 0].copy())
        ctx.save_for_backward(p1, p2, l1, l2, i)
        ctx.mark_non_differentiable(i)
        return d, i

    @staticmethod
    def backward(ctx, gd, _):
        p1, p2, l1, l2, i = ctx.saved_tensors
        g1, g2 = O.knn_points_bwd(p1.detach().numpy(), p2.detach().numpy(), l1.numpy(), l2.numpy(),
                                  i.numpy()[..., None], gd.contiguous().numpy()[..., None])
        return torch.from_numpy(g1), torch.from_numpy(g2), None, None


def _knn1_any_dtype(p1, p2, l1, l2):
    """The K = 1 search for the fp64 arbiter: the neighbour is chosen on the float32 values (the C oracle's search, i.e. the
    discrete decision of the fp32 paths), the squared distance to it is plain torch algebra in the inputs' own type."""
    _, i = O.knn_points(p1.detach().float().numpy(), p2.detach().float().numpy(), l1.numpy(), l2.numpy(), 1)
    i = torch.from_numpy(i[..., 0].copy())
    nb = p2.gather(1, i[..., None].expand(-1, -1, p2.shape[2]))
    d = (p1 - nb).square().sum(-1)
    valid = (torch.arange(p1.shape[1])[None] < l1[:, None]) & (l2[:, None] > 0)
    return torch.where(valid, d, torch.zeros((), dtype=d.dtype)), i


def _knn1_given(p1, p2, l1, l2, i):
    """The K = 1 'search' with the neighbour index GIVEN (routing-conditioned evaluation: the discrete decision of another
    implementation); the squared distance to it is plain torch algebra in the inputs' own type."""
    i = i.to(torch.int64)
    nb = p2.gather(1, i[..., None].expand(-1, -1, p2.shape[2]))
    d = (p1 - nb).square().sum(-1)
    valid = (torch.arange(p1.shape[1])[None] < l1[:, None]) & (l2[:, None] > 0)
    return torch.where(valid, d, torch.zeros((), dtype=d.dtype)), i


def chamfer_distance(x, y, padded=False, asymmetric=False, reverse_asymmetric=False, return_matching=False,
                     batch_reduction="mean", point_reduction="mean", nn_x=None, nn_y=None):
    """nn_x / nn_y: nearest-neighbour indices [B,P1] / [B,P2] of the x -> y / y -> x direction taken from another implementation
    (routing-conditioned evaluation); None: this evaluation's own search."""
    B, P1, D = x.shape
    P2 = y.shape[1]
    xl = torch.full((B,), P1, dtype=torch.int64)
    yl = torch.from_numpy(O.padded_lengths(y.detach().float().numpy())) if padded else torch.full((B,), P2, dtype=torch.int64)
    knn1 = _Knn1.apply if x.dtype == torch.float32 else _knn1_any_dtype
    y = y.to(x.dtype)
    cx, ix = knn1(x, y, xl, yl) if nn_x is None else _knn1_given(x, y, xl, yl, nn_x)
    cy, iy = knn1(y, x, yl, xl) if nn_y is None else _knn1_given(y, x, yl, xl, nn_y)
    # rows beyond the length already hold 0 (knn contract) == the reference's masking (:263-266)
    if point_reduction is not None:
        cx, cy = cx.sum(1), cy.sum(1)
        if point_reduction == "mean":
            cx, cy = cx / xl, cy / yl
    if batch_reduction is not None:
        cx, cy = cx.sum(), cy.sum()
        if batch_reduction == "mean":
            cx, cy = cx / B, cy / B
    d = cx if asymmetric else cy if reverse_asymmetric else cx + cy
    return (d, ix, iy) if return_matching else d


# ------------------------------------------------------------------------------------------------
# stroke-mask loss  (loss_handler.py:816-935) and the asymm_v6 total (:596-666)
# ------------------------------------------------------------------------------------------------
def stroke_masks_loss(idx_x, pred_masks, scores, stroke_ids, w_masks=1.0, w_conf=100.0, no_stroke_weight=1.0,
                      return_matching=False):
    B, M, S = pred_masks.shape
    tgt_ids = stroke_ids.gather(1, idx_x)
    assert not (tgt_ids == -1).any()
    matched_pred, matched_tgt, pairs = [], [], []
    tscore = torch.zeros(B, M, dtype=pred_masks.dtype)
    wts = torch.full((B, M), float(no_stroke_weight), dtype=pred_masks.dtype)
    for b in range(B):
        masks, _ = O.stroke_ids_to_masks(tgt_ids[b].float().numpy())
        cost = O.mask_bce_cost(pred_masks[b].detach().float().numpy(), masks)
        i, j = O.linear_sum_assignment(cost)
        pairs.append((i, j))
        matched_pred.append(pred_masks[b, torch.from_numpy(i)])
        matched_tgt.append(torch.from_numpy(masks[j]).to(pred_masks.dtype))
        tscore[b, torch.from_numpy(i)] = 1.0
        wts[b, torch.from_numpy(i)] = 1.0
    mp, mt = torch.cat(matched_pred), torch.cat(matched_tgt)
    mask_loss = F.binary_cross_entropy_with_logits(mp, mt, reduction="none").sum(-1).mean()
    conf_loss = F.binary_cross_entropy_with_logits(scores, tscore, weight=wts, reduction="none").mean()
    loss = w_masks * mask_loss + w_conf * conf_loss
    return (loss, pairs) if return_matching else loss


def asymm_v6_loss(y_pred, traj, pred_masks, scores, stroke_ids, traj_as_pc, cfg, nn_routes=None, nn_out=None):
    """cfg: dict with the weights read at loss_handler.py:660-664,934.
    nn_routes: [pred segment -> GT segment, GT point -> predicted pose, GT segment -> pred segment] nearest-neighbour indices of the
    three terms taken from another implementation (routing-conditioned evaluation); nn_out: list receiving this evaluation's own."""
    B = y_pred.shape[0]
    r = nn_routes if nn_routes is not None else (None, None, None)
    d1, idx_x, _ = chamfer_distance(y_pred, traj, padded=True, asymmetric=True, return_matching=True,
                                    point_reduction=None, batch_reduction=None, nn_x=r[0])
    t1 = 100 * d1.mean()
    t2, _, i2 = chamfer_distance(y_pred.reshape(B, -1, 6), traj_as_pc, padded=True, reverse_asymmetric=True, return_matching=True, nn_y=r[1])
    t3, _, i3 = chamfer_distance(y_pred, traj, padded=True, reverse_asymmetric=True, return_matching=True, nn_y=r[2])
    t2, t3 = 100 * t2, 100 * t3
    if nn_out is not None:
        nn_out.extend([idx_x, i2, i3])
    t4 = stroke_masks_loss(idx_x, pred_masks, scores, stroke_ids, cfg["explicit_weight_stroke_masks"],
                           cfg["explicit_weight_stroke_masks_confidence"], cfg["explicit_no_stroke_weight"])
    return (cfg["weight_asymm_segment_chamfer"] * t1 + cfg["weight_reverse_asymm_point_chamfer"] * t2
            + cfg["weight_reverse_asymm_segment_chamfer"] * t3 + t4)


def hungarian_match(outputs, targets):
    """models/hungarianMatcher.py:31-63 without the cross-batch cdist waste."""
    res = []
    for b, t in enumerate(targets):
        c = O.cdist(outputs[b].detach().numpy(), t.detach().numpy())
        res.append(O.linear_sum_assignment(c))
    return res
