"""MaskPlanner backbones on top of the MI355X set-abstraction stack.

Mirrors `models/pointnet2_cls_ssg.py` of the reference -- all six classes `models/__init__.py:20` imports
(PointNet2Regressor :12, _SoPs :85, _3Dbbox :177, _StrokeMasks :233, _StrokeMasks_RetroCompatible :348, _StrokeWise :463):
same class names, constructor arguments, forward signature/outputs and -- for checkpoint compatibility (test_maskplanner.py:162-188) -- identical `state_dict`
keys and shapes: sa{1,2,3}.mlp_convs/mlp_bns.*, fc1/fc2/fc3, bn1/bn2, fc_normals, sm_fc1..3, sm_bn1/2,
mask_conf_out, seg_conf_fc1/2, seg_conf_out.  The encoder (`sa1..sa3`) is the hot path and runs the HIP
kernels; the regression heads are a handful of dense layers on a [B,1024] feature and stay on rocBLAS via torch
(weight-bandwidth bound GEMMs, "next" in the scope table).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import graphed, ops, sa_mlp
from .factor_heads import factor_linear, factor_linear2, head_block, head_block_ok, head_blocks2, head_blocks2_ok

SAMPLE_AHEAD = True      # False (tests): every level samples in line
from .pointnet2_utils import PointNetSetAbstraction, PointNetSetAbstractionMsg


class _SSGEncoder(nn.Module):
    """sa1 (512 centroids, r=.2, K=32) -> sa2 (128, r=.4, K=64) -> sa3 (group all) -> [B,1024]
    (models/pointnet2_cls_ssg.py:37-39, 266-268)."""

    def zero_grad(self, set_to_none=True):
        """nn.Module.zero_grad through the flattened parameter list (graphed.fast_zero_grad): the reference's loop calls it twice per iteration
        (train_maskplanner.py:183, 226), 0.1 ms of host time each with the device idle."""
        from . import graphed
        graphed.fast_zero_grad(self, set_to_none)


    def _build_encoder(self, normal_channel, inputdim):
        in_channel = 6 if normal_channel else 3
        if inputdim is not None:
            in_channel = inputdim
        self.normal_channel = normal_channel
        self.sa1 = PointNetSetAbstraction(npoint=512, radius=0.2, nsample=32, in_channel=in_channel,
                                          mlp=[64, 64, 128], group_all=False)
        self.sa2 = PointNetSetAbstraction(npoint=128, radius=0.4, nsample=64, in_channel=128 + 3,
                                          mlp=[128, 128, 256], group_all=False)
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=256 + 3,
                                          mlp=[256, 512, 1024], group_all=True)

    def encode(self, xyz):
        """xyz [B,3(+3),N] -> global feature [B,1024]."""
        B = xyz.shape[0]
        norm = None
        if self.normal_channel:
            norm = xyz[:, 3:, :]
            xyz = xyz[:, :3, :]
        if xyz.is_cuda and all(isinstance(m, PointNetSetAbstraction) for m in (self.sa1, self.sa2, self.sa3)):
            # the three first-layer weights into the fused MLP's column order with one launch (and one for their gradients)
            sa_mlp.prepermute([(self.sa1.mlp_convs[0], "feats_first" if norm is not None else "xyz_first"),
                               (self.sa2.mlp_convs[0], "feats_first"), (self.sa3.mlp_convs[0], "feats_first")])
        self._sample_ahead(xyz)
        l1_xyz, l1_points = self.sa1(xyz, norm)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points)
        _, l3_points = self.sa3(l2_xyz, l2_points)
        return l3_points.reshape(B, 1024)

    def _sample_ahead(self, xyz):
        """Eager calls without a supplied sampling plan (an unchanged training loop): the second level samples nothing but the first
        level's centroids, which exist as soon as the first level's FPS is done -- so both levels are sampled here, the second one on
        a side stream underneath the first level's grouping + MLP (its FPS is a one-wave-per-cloud latency chain: 60 + 40 us that
        otherwise sit on the critical path).  The FPS starts are drawn in the reference's order; the levels pick the plans up through
        pointnet2_utils' plan queue."""
        from . import pointnet2_utils as pu
        sa1, sa2 = self.sa1, self.sa2
        # (while a harness records its own graphs the plan is supplied from outside; graphed.py records THIS fork and join: FORK_IN_CAPTURE)
        if (not SAMPLE_AHEAD or not xyz.is_cuda or (torch.cuda.is_current_stream_capturing() and not (graphed.FORK_IN_CAPTURE and pu._capture_starts is not None))
                or getattr(sa1, "group_all", True)
                or getattr(sa2, "group_all", True) or not isinstance(sa1, PointNetSetAbstraction) or not isinstance(sa2, PointNetSetAbstraction)):
            return
        pm = pu._points_major(xyz)                     # what sa1.forward computes (the same storage for a permuted [B,N,3] input)
        if pm.data_ptr() != pu._points_major(xyz).data_ptr() or pu.has_prefetched(pm, sa1.npoint, sa1.radius, sa1.nsample):
            return                                     # (a fresh copy per call: the plan could not be found again) / a plan is there already
        B, N, _ = pm.shape
        dev = pm.device
        main = torch.cuda.current_stream(dev)
        s1 = pu._draw_fps_start(B, N, dev)
        s2 = pu._draw_fps_start(B, sa1.npoint, dev)
        fps1, new1 = ops.fps(pm, sa1.npoint, s1, return_xyz=True)
        side = pu._side_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fps2, new2 = ops.fps(new1, sa2.npoint, s2, return_xyz=True)
            idx2 = ops.ball_query(sa2.radius, sa2.nsample, new1, new2)
            ev = torch.cuda.Event()
            ev.record(side)
        # lifetimes across the two streams: s2 and new1 were allocated on the step's stream and are READ on the side stream
        # (the allocator must not hand their blocks out again before the side stream's FPS has run); the three outputs were
        # allocated on the side stream and are read by the step's stream
        for t in (s2, new1):
            t.record_stream(side)
        for t in (fps2, new2, idx2):
            t.record_stream(main)
        idx1 = ops.ball_query(sa1.radius, sa1.nsample, pm, new1)
        pu.supply_sampling(pm, sa1.npoint, sa1.radius, sa1.nsample, (fps1, new1, idx1))
        pu._prefetched.setdefault((new1.data_ptr(), sa2.npoint, float(sa2.radius), sa2.nsample), []).append((ev, fps2, new2, idx2))


def _block(model, lin_out, bn, layer):
    """self.dropout(F.relu(bn(.))) of one head block (:309-327).  With `model.fused_dropout` set (a device int64 (seed, step) tensor, see
    harness.TrainStep) the train-mode dropout rides in the BatchNorm + ReLU launch (ops.bn_relu_rows(dropout=...)): its own
    counter-based mask instead of torch's Philox stream, no kernel and no saved mask of its own."""
    rng = getattr(model, "fused_dropout", None)
    sync = getattr(bn, "sync_bn", None)
    if (rng is not None and model.training and bn.training and lin_out.is_cuda and lin_out.dtype == torch.float32 and lin_out.ndim == 2
            and (sync is None or sync is False)):
        return ops.bn_relu_rows(lin_out, bn, dropout=(model.dropout.p, rng, layer))
    return model.dropout(_bn_relu(lin_out, bn))


HEAD_BLOCK = os.environ.get("MASKPLANNER_HEAD_BLOCK", "1") != "0"    # (A/B switch while the block kernels are new) False: Linear + ops.bn_relu_rows launches


def _head_block(model, x, linear, bn, store, key, layer):
    """One block of the heads: dropout(relu(bn(linear(x)))) -- one launch each way where csrc/head_linear.hip applies
    (factor_heads.head_block), else Linear, then _block."""
    if HEAD_BLOCK and head_block_ok(x, linear, bn):
        rng = getattr(model, "fused_dropout", None)
        if rng is not None and model.training and bn.training:
            return head_block(x, linear, bn, store, key, dropout=(model.dropout.p, rng, layer))
        return model.dropout(head_block(x, linear, bn, store, key))
    return _block(model, factor_linear(x, linear, store, key), bn, layer)


def _bn_relu(x, bn):
    """F.relu(bn(x)) of the head blocks (:309-327); on the GPU one fused launch (ops.bn_relu_rows) instead of BatchNorm's
    statistics / transform / running-stat kernels + clamp.  The caller has advanced num_batches_tracked (_tick)."""
    sync = getattr(bn, "sync_bn", None)
    if sync is not None and bn.training:
        from . import sync_bn
        group = sync_bn.resolve(sync)
        if group is not False:
            if not x.is_cuda:     # (on the GPU the caller advanced the counters of all its BatchNorm layers in one launch: _tick)
                _tick(bn)
            return sync_bn.bn_relu_rows_sync(x, bn, group)
    if x.is_cuda and x.dtype == torch.float32 and x.ndim == 2:
        return ops.bn_relu_rows(x, bn)
    return F.relu(bn(x))


def _tick(*bns):
    """num_batches_tracked += 1 for train-mode BatchNorm layers, one launch for all of them (nn.BatchNorm does it per layer)."""
    counters = [bn.num_batches_tracked for bn in bns if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None]
    if counters:
        from . import sa_mlp
        if sa_mlp.DEFERRED_TICKS is not None:
            sa_mlp.DEFERRED_TICKS.extend(counters)
        else:
            torch._foreach_add_(counters, 1)


def _pose_output(x, normals_raw, B, out_vectors, weight_orient):
    """cat(position, weight_orient * unit normal) per pose, poses interleaved per output vector (:332-339)."""
    if x.is_cuda and x.dtype == torch.float32 and x.shape == normals_raw.shape:
        return ops.pose_output(x, normals_raw, weight_orient).view(B, out_vectors, -1)       # one launch (and one backward)
    normals = F.normalize(torch.tanh(normals_raw).view(B, -1, 3), dim=-1) * weight_orient
    return torch.cat((x.view(B, -1, 3), normals), dim=-1).view(B, out_vectors, -1)


class PointNet2Regressor(_SSGEncoder):
    """models/pointnet2_cls_ssg.py:12-81 (`backbone: pointnet2`)."""

    def __init__(self, outdim=3, outdim_orient=3, weight_orient=1., normal_channel=False, out_vectors=1500,
                 hidden_size=(1024, 1024), inputdim=None):
        super().__init__()
        self.outdim, self.outdim_orient = outdim, outdim_orient
        self.out_vectors, self.weight_orient = out_vectors, weight_orient
        self._build_encoder(normal_channel, inputdim)
        self.fc1 = nn.Linear(1024, hidden_size[0])
        self.fc2 = nn.Linear(hidden_size[0], hidden_size[1])
        self.fc3 = nn.Linear(hidden_size[1], out_vectors * outdim)
        if outdim_orient > 0:
            self.fc_normals = nn.Linear(hidden_size[1], out_vectors * outdim_orient)
            self.tanh = nn.Tanh()
        self.dropout = nn.Dropout(p=0.3)
        self.bn1 = nn.BatchNorm1d(hidden_size[0])
        self.bn2 = nn.BatchNorm1d(hidden_size[1])

    def forward(self, xyz):
        B = xyz.shape[0]
        feat = self.encode(xyz)
        fused = feat.is_cuda
        if fused:
            _tick(self.bn1, self.bn2)
        act = _bn_relu
        x = self.dropout(act(self.fc1(feat), self.bn1))
        final = self.dropout(act(self.fc2(x), self.bn2))
        x = self.fc3(final)
        if self.outdim_orient > 0:
            return _pose_output(x, self.fc_normals(final), B, self.out_vectors, self.weight_orient)
        return x.view(B, self.out_vectors, self.outdim)


class PointNet2Regressor_StrokeMasks(_SSGEncoder):
    """models/pointnet2_cls_ssg.py:233-344 (`backbone: pointnet2_strokemasks`, the MaskPlanner model).
    forward(xyz [B,3,N]) -> (out [B,S,D], sm_out [B,M,S] | None, mask_conf [B,M] | None, seg_conf | None)."""

    def __init__(self, outdim=3, outdim_orient=3, weight_orient=1., normal_channel=False, out_vectors=1500,
                 hidden_size=(1024, 1024), inputdim=None, pred_stroke_masks=False, n_stroke_masks=None,
                 mask_confidence_scores=False, segment_confidence_scores=False):
        super().__init__()
        self.outdim, self.outdim_orient = outdim, outdim_orient
        self.out_vectors, self.weight_orient = out_vectors, weight_orient
        self.pred_stroke_masks, self.n_stroke_masks = pred_stroke_masks, n_stroke_masks
        self.mask_confidence_scores = mask_confidence_scores
        self.segment_confidence_scores = segment_confidence_scores
        self._build_encoder(normal_channel, inputdim)
        h0, h1 = hidden_size
        self.fc1 = nn.Linear(1024, h0)
        self.fc2 = nn.Linear(h0, h1)
        self.fc3 = nn.Linear(h1, out_vectors * outdim)
        self.dropout = nn.Dropout(p=0.3)
        self.bn1 = nn.BatchNorm1d(h0)
        self.bn2 = nn.BatchNorm1d(h1)
        if outdim_orient > 0:
            self.fc_normals = nn.Linear(h1, out_vectors * outdim_orient)
            self.tanh = nn.Tanh()
        if segment_confidence_scores:
            self.seg_conf_fc1 = nn.Linear(1024, h0)
            self.seg_conf_fc2 = nn.Linear(h0, h1)
            self.seg_conf_out = nn.Linear(h1, out_vectors)
        if pred_stroke_masks:
            self.sm_fc1 = nn.Linear(1024, h0)
            self.sm_fc2 = nn.Linear(h0, h1)
            self.sm_fc3 = nn.Linear(h1, out_vectors * n_stroke_masks)
            self.sm_bn1 = nn.BatchNorm1d(h0)
            self.sm_bn2 = nn.BatchNorm1d(h1)
            if mask_confidence_scores:
                setattr(self, self._CONF_LAYER, nn.Linear(h1, n_stroke_masks))

    _CONF_LAYER = "mask_conf_out"      # `out_confidence` in checkpoints older than the rename (RetroCompatible, :410)

    # set to a dict by a training harness that uses factor_heads.FactorAdam: the head matrices fed by a [B, 1024] feature
    # (fc1/fc2/fc3/fc_normals and the sm_ twins) then keep their gradient as rank-B factors instead of materialising dW
    # (default None: plain nn.Linear behaviour)
    factor_store = None
    # set to a device int64 tensor (seed, step) by a training harness: the four train-mode dropouts of the heads are then applied inside
    # the BatchNorm + ReLU launches (_block); the harness advances the step counter once per step (default None: nn.Dropout)
    fused_dropout = None

    def forward(self, xyz):
        # [r5] inside an unchanged training loop: the whole call (and its backward) replayed from two recorded graphs (graphed.py)
        out = graphed.call(self, xyz)
        return out if out is not None else self._forward_eager(xyz)

    def _forward_eager(self, xyz):
        return self.heads(self.encode(xyz))

    def _apply(self, fn, *a, **kw):
        graphed.reset(self)          # (.to() / .half() / ... replace the parameters the graphs were recorded with)
        return super()._apply(fn, *a, **kw)

    def heads(self, feat):
        """Global feature [B,1024] -> (out, sm_out, mask_conf, seg_conf): everything behind the encoder (:309-341).  Split from
        forward() so that a harness can schedule the two halves separately (harness.TrainStep records them as two graphs)."""
        B = feat.shape[0]
        fs = self.factor_store
        fused = feat.is_cuda
        if fused:
            _tick(*([self.bn1, self.bn2] + ([self.sm_bn1, self.sm_bn2] if self.pred_stroke_masks else [])))
        # [r4] the pose branch and the stroke-mask branch advance side by side: fc1 / sm_fc1 (both on the global feature) and fc2 / sm_fc2
        # are one launch each way per pair (factor_heads.head_blocks2); the statements below keep the reference's order otherwise
        # (with torch's own nn.Dropout active -- a training model without `fused_dropout`, i.e. the drop-in model inside the reference's loop --
        # the pairs would draw their masks in the order fc1, sm_fc1, fc2, sm_fc2 where the reference draws fc1, fc2, [seg_conf x 2], sm_fc1,
        # sm_fc2 (:309-324): that path keeps the reference's statement order, so that one torch seed gives the reference's masks)
        torch_dropout = self.training and self.dropout.p > 0 and getattr(self, "fused_dropout", None) is None
        paired = (HEAD_BLOCK and self.pred_stroke_masks and not torch_dropout
                  and head_blocks2_ok(feat, None, self.fc1, self.bn1, self.sm_fc1, self.sm_bn1)
                  and self.fc2.weight.shape == self.sm_fc2.weight.shape and self.fc1.weight.shape[0] == self.sm_fc1.weight.shape[0])
        s2 = None
        if paired:
            rng = getattr(self, "fused_dropout", None)
            drop = (self.dropout.p, rng) if (rng is not None and self.training and self.bn1.training) else None
            post = (lambda t: t) if drop is not None else self.dropout
            x, s1 = head_blocks2(feat, None, self.fc1, self.bn1, self.sm_fc1, self.sm_bn1, fs, "fc1.weight", "sm_fc1.weight", drop, (0, 2))
            x, s1 = post(x), post(s1)
            if head_blocks2_ok(x, s1, self.fc2, self.bn2, self.sm_fc2, self.sm_bn2):
                final, s2 = head_blocks2(x, s1, self.fc2, self.bn2, self.sm_fc2, self.sm_bn2, fs, "fc2.weight", "sm_fc2.weight", drop, (1, 3))
                final, s2 = post(final), post(s2)
            else:
                final = _head_block(self, x, self.fc2, self.bn2, fs, "fc2.weight", 1)
        else:
            x = _head_block(self, feat, self.fc1, self.bn1, fs, "fc1.weight", 0)
            final = _head_block(self, x, self.fc2, self.bn2, fs, "fc2.weight", 1)
        # (the order of the launches follows the reference's statements; fc3 and fc_normals read the same activation: one launch)
        if self.outdim_orient > 0:
            x, raw_normals = factor_linear2(final, self.fc3, self.fc_normals, fs, "fc3.weight", "fc_normals.weight")
        else:
            x = factor_linear(final, self.fc3, fs, "fc3.weight")

        seg_conf = None
        if self.segment_confidence_scores:
            s = self.dropout(F.relu(self.seg_conf_fc1(feat)))
            s = self.dropout(F.relu(self.seg_conf_fc2(s)))
            seg_conf = torch.sigmoid(self.seg_conf_out(s))

        sm_out, mask_conf = None, None
        if self.pred_stroke_masks:
            if not paired:
                s1 = _head_block(self, feat, self.sm_fc1, self.sm_bn1, fs, "sm_fc1.weight", 2)
            if s2 is None:
                s2 = _head_block(self, s1, self.sm_fc2, self.sm_bn2, fs, "sm_fc2.weight", 3)
            if self.mask_confidence_scores:     # (sm_fc3 and the confidence layer read the same activation: one launch each way)
                sm_out, mask_conf = factor_linear2(s2, self.sm_fc3, getattr(self, self._CONF_LAYER), fs, "sm_fc3.weight", None)
            else:
                sm_out = factor_linear(s2, self.sm_fc3, fs, "sm_fc3.weight")
            sm_out = sm_out.view(B, self.n_stroke_masks, -1)

        if self.outdim_orient > 0:
            out = _pose_output(x, raw_normals, B, self.out_vectors, self.weight_orient)
        else:
            out = x.view(B, self.out_vectors, self.outdim)
        return out, sm_out, mask_conf, seg_conf


class PointNet2Regressor_StrokeMasks_RetroCompatible(PointNet2Regressor_StrokeMasks):
    """models/pointnet2_cls_ssg.py:348-459 (`backbone: pointnet2_strokemasks_retrocompatible`): the same network with the
    mask-confidence layer under its old state_dict name `out_confidence` (test_maskplanner.py:180-189 falls back to it for
    checkpoints written before the rename)."""
    _CONF_LAYER = "out_confidence"


class _TrunkRegressor(_SSGEncoder):
    """What the remaining siblings share (:116-131, :197-207, :506-518): the SSG encoder and the fc1/bn1 -> fc2/bn2 -> fc3
    trunk on the global feature.  Registration order == the reference's, so state_dict keys come out in the same order."""

    def _build_trunk(self, normal_channel, inputdim, hidden_size, n_out, n_out_orient=0):
        self._build_encoder(normal_channel, inputdim)
        self.fc1 = nn.Linear(1024, hidden_size[0])
        self.fc2 = nn.Linear(hidden_size[0], hidden_size[1])
        self.fc3 = nn.Linear(hidden_size[1], n_out)
        self.dropout = nn.Dropout(p=0.3)
        self.bn1 = nn.BatchNorm1d(hidden_size[0])
        self.bn2 = nn.BatchNorm1d(hidden_size[1])
        if n_out_orient > 0:
            self.fc_normals = nn.Linear(hidden_size[1], n_out_orient)
            self.tanh = nn.Tanh()

    def _trunk(self, xyz):
        """-> (global feature [B,1024], last hidden activation [B,h1], fc3 output)."""
        feat = self.encode(xyz)
        fused = feat.is_cuda
        if fused:
            _tick(self.bn1, self.bn2)
        act = _bn_relu
        last = self.dropout(act(self.fc2(self.dropout(act(self.fc1(feat), self.bn1))), self.bn2))
        return feat, last, self.fc3(last)

    def _poses(self, x, last, B):
        if self.outdim_orient > 0:
            return _pose_output(x, self.fc_normals(last), B, self.out_vectors, self.weight_orient)
        return x.view(B, self.out_vectors, self.outdim)


class PointNet2Regressor_SoPs(_TrunkRegressor):
    """models/pointnet2_cls_ssg.py:85-174 (`backbone: pointnet2_sops`): start-of-path poses + optional per-pose confidence.
    forward(xyz, return_object_features=False) -> (out [B,V,D], sop_conf [B,V] | None[, global feature])."""

    def __init__(self, out_vectors=10, outdim=3, outdim_orient=3, weight_orient=1., normal_channel=False,
                 hidden_size=(1024, 1024), inputdim=None, sop_confidence_scores=False):
        super().__init__()
        self.outdim, self.outdim_orient = outdim, outdim_orient
        self.out_vectors, self.weight_orient = out_vectors, weight_orient
        self.sop_confidence_scores = sop_confidence_scores
        self._build_trunk(normal_channel, inputdim, hidden_size, out_vectors * outdim, out_vectors * outdim_orient)
        if sop_confidence_scores:
            self.sop_conf_out = nn.Linear(hidden_size[1], out_vectors)

    def forward(self, xyz, return_object_features=False):
        feat, last, x = self._trunk(xyz)
        conf = self.sop_conf_out(last) if self.sop_confidence_scores else None
        out = self._poses(x, last, xyz.shape[0])
        return (out, conf, feat) if return_object_features else (out, conf)


class PointNet2Regressor_3Dbbox(_TrunkRegressor):
    """models/pointnet2_cls_ssg.py:177-229 (`backbone: pointnet2_3dbbox`): out_bboxes boxes as (centre xyz, size whd)."""

    def __init__(self, out_bboxes=10, normal_channel=False, hidden_size=(1024, 1024), inputdim=None):
        super().__init__()
        self.out_bboxes, self.outdim = out_bboxes, 6
        self._build_trunk(normal_channel, inputdim, hidden_size, out_bboxes * 6)

    def forward(self, xyz):
        return self._trunk(xyz)[2].view(xyz.shape[0], self.out_bboxes, self.outdim)


class PointNet2Regressor_StrokeWise(_TrunkRegressor):
    """models/pointnet2_cls_ssg.py:463-556 (`backbone: pointnet2_strokewise`): one output vector per stroke + optional
    per-stroke and per-point confidences.  forward -> (out [B,V,D], point_conf [B,V,n] | None, stroke_conf [B,V] | None).
    (With point_confidence_scores=False the reference's forward stops on an unbound local at :556; None is returned here.)"""

    def __init__(self, outdim=3, outdim_orient=3, weight_orient=1., normal_channel=False, out_vectors=1500,
                 hidden_size=(1024, 1024), inputdim=None, stroke_confidence_scores=False, point_confidence_scores=False,
                 n_points_per_out_vector=None):
        super().__init__()
        self.outdim, self.outdim_orient = outdim, outdim_orient
        self.out_vectors, self.weight_orient = out_vectors, weight_orient
        self.stroke_confidence_scores, self.point_confidence_scores = stroke_confidence_scores, point_confidence_scores
        self.n_points_per_out_vector = n_points_per_out_vector
        self._build_trunk(normal_channel, inputdim, hidden_size, out_vectors * outdim, out_vectors * outdim_orient)
        if stroke_confidence_scores:
            self.stroke_conf_out = nn.Linear(hidden_size[1], out_vectors)
        if point_confidence_scores:
            self.point_conf_out = nn.Linear(hidden_size[1], out_vectors * n_points_per_out_vector)

    def forward(self, xyz):
        B = xyz.shape[0]
        _, last, x = self._trunk(xyz)
        stroke_conf = self.stroke_conf_out(last) if self.stroke_confidence_scores else None
        point_conf = None
        if self.point_confidence_scores:
            point_conf = self.point_conf_out(last).view(B, self.out_vectors, self.n_points_per_out_vector)
        return self._poses(x, last, B), point_conf, stroke_conf


class PointNet2Regressor_StrokeMasks_MSG(PointNet2Regressor_StrokeMasks):
    """The MaskPlanner heads on a multi-scale-grouping encoder (BASELINE configs[4]: "MSG encoder (multi-radius ball-query)").
    The reference ships the layer -- `PointNetSetAbstractionMsg`, models/pointnet2_utils.py:219-276 -- but no model that
    stacks it (models/__init__.py:66-217 has no MSG backbone), so the stack follows the layer's upstream home
    (yanx27 pointnet2_cls_msg): 512 centroids x radii (.1, .2, .4) x (16, 32, 128) neighbours -> 320 channels, 128 centroids x
    (.2, .4, .8) x (32, 64, 128) -> 640 channels, then the reference's group_all level (643 -> 256 -> 512 -> 1024).  Heads,
    outputs and head state_dict keys are PointNet2Regressor_StrokeMasks'."""
    MSG1 = dict(npoint=512, radius_list=[0.1, 0.2, 0.4], nsample_list=[16, 32, 128],
                mlp_list=[[32, 32, 64], [64, 64, 128], [64, 96, 128]])
    MSG2 = dict(npoint=128, radius_list=[0.2, 0.4, 0.8], nsample_list=[32, 64, 128],
                mlp_list=[[64, 64, 128], [128, 128, 256], [128, 128, 256]])

    def _build_encoder(self, normal_channel, inputdim):
        if normal_channel or inputdim not in (None, 3):
            raise NotImplementedError("the MSG encoder of this build takes bare coordinates")
        self.normal_channel = False
        self.sa1 = PointNetSetAbstractionMsg(in_channel=0, **self.MSG1)
        self.sa2 = PointNetSetAbstractionMsg(in_channel=320, **self.MSG2)
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=640 + 3,
                                          mlp=[256, 512, 1024], group_all=True)

    def encode(self, xyz):
        B = xyz.shape[0]
        if xyz.is_cuda:
            # the first-layer weights that need the kernels' column order (three 3 -> 4 pads of the first level, the 643-column
            # group_all level) with one launch, and one for their gradients, instead of eight
            sa_mlp.prepermute([(convs[0], "xyz_first") for convs in self.sa1.conv_blocks] + [(self.sa3.mlp_convs[0], "feats_first")])
        l1_xyz, l1_points = self.sa1(xyz, None)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points)
        _, l3_points = self.sa3(l2_xyz, l2_points)
        return l3_points.reshape(B, 1024)


def set_mlp_dtype(model, dtype):
    """Operand type ("f32" | "bf16") of the grouped-MLP contractions of every set-abstraction level of `model`."""
    if dtype not in ("f32", "bf16"):
        raise ValueError("dtype must be 'f32' or 'bf16'")
    for m in model.modules():
        if isinstance(m, (PointNetSetAbstraction, PointNetSetAbstractionMsg)):
            m.mlp_dtype = dtype
    return model


def maskplanner_model(category, lambda_points=4, overlapping=1, outdim=6, orient_outdim=3, weight_orient=0.25,
                      hidden_size=(1024, 1024), encoder="ssg", mlp_dtype="f32"):
    """The model `get_model(config, which='pointnet2_strokemasks', io_type='MaskPlanner')` builds
    (models/__init__.py:111-122, 295-318) for a synthetic.Category.  encoder="msg": the same heads on the multi-scale
    encoder (PointNet2Regressor_StrokeMasks_MSG); mlp_dtype: see set_mlp_dtype."""
    if encoder not in ("ssg", "msg"):
        raise ValueError("encoder must be 'ssg' or 'msg'")
    cls = PointNet2Regressor_StrokeMasks if encoder == "ssg" else PointNet2Regressor_StrokeMasks_MSG
    return set_mlp_dtype(_build_maskplanner(cls, category, lambda_points, outdim, orient_outdim, weight_orient, hidden_size), mlp_dtype)


def _build_maskplanner(cls, category, lambda_points, outdim, orient_outdim, weight_orient, hidden_size):
    return cls(
        out_vectors=category.out_vectors, outdim=(outdim - orient_outdim) * lambda_points,
        outdim_orient=orient_outdim * lambda_points, weight_orient=weight_orient, hidden_size=hidden_size,
        pred_stroke_masks=True, n_stroke_masks=category.max_n_strokes, mask_confidence_scores=True,
        segment_confidence_scores=False)
