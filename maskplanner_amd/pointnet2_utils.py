"""Drop-in for the reference's `models/pointnet2_utils.py` on MI355X.

Same module-level names, signatures, return shapes/dtypes and `state_dict` layout as the reference
(models/pointnet2_utils.py:21-276), so `models/pointnet2_cls_ssg.py` / `pointnet2_seg.py` -- and through them the
unchanged `train_maskplanner.py` / `test_maskplanner.py` -- can import this module in its place.  Every function
runs hand-written gfx950 kernels through libmaskplanner_hip.so (see ops.py); nothing falls back to the reference's
tensor algebra, and CPU tensors are refused.

Differences that are NOT visible in results:
  * farthest_point_sample: one kernel with the cloud resident on chip instead of ~8 launches + a host sync per
    step (:79-85).  The start index is still drawn with torch.randint on the global CPU generator (:77), in the
    same call order, so seeded runs pick the same starts.
  * query_ball_point: an index-ordered scan with early exit instead of materialising and sorting [B,S,N] (:102-105).
  * the set-abstraction MLP works on positions-major activations [B*S*K, C] (1x1 Conv2d == a GEMM over the
    channel axis; BatchNorm2d statistics == statistics over all rows).
"""
import contextlib
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from . import sa_mlp

_fps_start_queue = []
_ZEROS = {}


def _zeros(shape, device):
    """A cached all-zero fp32 tensor (read-only by convention): the group-all level needs `new_xyz = 0` and a zero padding
    column every step; a fill launch each would cost more than the data."""
    key = (tuple(shape), device)
    t = _ZEROS.get(key)
    if t is None:
        t = _ZEROS[key] = torch.zeros(shape, dtype=torch.float32, device=device)
    return t


@contextlib.contextmanager
def fps_start_override(starts):
    """Inject explicit FPS start indices (one [B] tensor per upcoming farthest_point_sample call, in call
    order) instead of drawing them -- parity tests and benchmarks use it for reproducibility."""
    _fps_start_queue.extend(starts)
    try:
        yield
    finally:
        del _fps_start_queue[:]


# ---- sampling prefetch ----------------------------------------------------------------------------------------
# FPS is a chain of dependent steps that can use only one workgroup per cloud (32 of 256 CUs at B=32) and depends on
# nothing but the input cloud.  A training loop that already holds the NEXT batch can therefore run its first-level
# sampling (FPS + ball query) on a side HIP stream underneath the current step's backward pass.
# [r1 measurement: 5.77 -> 5.71 ms/step at B=32 -- the side stream's 61 KB-LDS workgroups barely get scheduled next to
#  full-chip GEMM grids, so the harness leaves it off by default.]
_prefetched = {}
_side_streams = {}


def _side_stream(dev):
    side = _side_streams.get(dev)
    if side is None:
        side = _side_streams[dev] = torch.cuda.Stream(device=dev)
    return side


def prefetch_sampling(xyz, npoint, radius, nsample, fps_start):
    """Queue first-level sampling of `xyz` [B,N,3] (points-major, contiguous) on a side stream.  The next
    sample_and_group() call with the same tensor and parameters picks the result up instead of recomputing it."""
    dev = xyz.device
    main = torch.cuda.current_stream(dev)
    side = _side_streams.get(dev)
    if side is None:
        side = _side_streams[dev] = torch.cuda.Stream(device=dev)
    side.wait_stream(main)  # the cloud must be complete before the side stream reads it
    with torch.cuda.stream(side):
        start = torch.as_tensor(fps_start, dtype=torch.long).to(dev)
        fps_idx, new_xyz = ops.fps(xyz, npoint, start, return_xyz=True)
        idx = ops.ball_query(radius, nsample, xyz, new_xyz)
        ev = torch.cuda.Event()
        ev.record(side)
    for t in (fps_idx, new_xyz, idx):
        t.record_stream(main)  # allocated on the side stream, consumed on the main one
    _prefetched.setdefault((xyz.data_ptr(), npoint, float(radius), nsample), []).append((ev, fps_idx, new_xyz, idx))


def _plan_key(xyz, npoint, radius, nsample):
    """Plans are keyed by the cloud's storage and the sampling parameters; a multi-scale level has a tuple of radii / group sizes."""
    r = tuple(float(x) for x in radius) if isinstance(radius, (list, tuple)) else float(radius)
    k = tuple(int(x) for x in nsample) if isinstance(nsample, (list, tuple)) else nsample
    return (xyz.data_ptr(), npoint, r, k)


# idx.data_ptr() -> the grouped, centred coordinate rows [B, S, K, 4] of a level WITHOUT input features (ops.group(xyz, None, new_xyz,
# idx, pad_to=4)): they depend on the cloud and its sampling plan alone, so a harness computes them with the plan, off the step's stream
_grouped_xyz = {}


def supply_sampling(xyz, npoint, radius, nsample, plan):
    """Hand sample_and_group() a finished first-level sampling (fps_idx, new_xyz, idx) of `xyz`, already ordered on the
    current stream (the pipelined step of harness.TrainStep: the plan was computed during the previous step).  A multi-scale
    level (radius / nsample: the lists of PointNetSetAbstractionMsg) takes idx as a tuple, one ball query per radius."""
    _prefetched.setdefault(_plan_key(xyz, npoint, radius, nsample), []).append((None,) + tuple(plan))


def clear_prefetched():
    """Drop every queued sampling plan.  Plans are keyed by the cloud's storage address: one left behind by an aborted or
    skipped step could otherwise be consumed by a different cloud that the allocator later placed at the same address."""
    _prefetched.clear()


def has_prefetched(xyz, npoint, radius, nsample):
    return bool(_prefetched.get(_plan_key(xyz, npoint, radius, nsample)))


def _take_prefetched(xyz, npoint, radius, nsample):
    key = _plan_key(xyz, npoint, radius, nsample)
    q = _prefetched.get(key)
    if not q:
        return None
    ev, fps_idx, new_xyz, idx = q.pop(0)
    if not q:
        del _prefetched[key]
    if ev is not None:
        torch.cuda.current_stream(xyz.device).wait_event(ev)
    return fps_idx, new_xyz, idx


def square_distance(src, dst):
    """[B,N,C] x [B,M,C] -> [B,N,M] squared distances in the reference's expanded form (:21-42)."""
    if src.shape[-1] == 3:
        return ops.square_distance(src, dst)
    raise NotImplementedError("square_distance: only 3-D points are on the MaskPlanner hot path")


def index_points(points, idx):
    """points [B,N,C], idx [B,S] or [B,S,K] -> points[b, idx[b,...], :] (:45-62)."""
    return ops.index_points(points, idx)


_capture_starts = None      # graphed.py, while it records a forward: [(static tensor, N)] handed out in call order instead of fresh draws --
_draw_log = None            # allocated BEFORE the recording from the (B, N) of the draws an eager forward made (logged here)


def _draw_fps_start(B, N, device):
    if _capture_starts is not None:
        if not _capture_starts or tuple(_capture_starts[0][0].shape) != (B,) or _capture_starts[0][1] != N:
            raise RuntimeError("the recorded forward draws other FPS starts than the eager forward before it did")
        return _capture_starts.pop(0)[0]
    if _draw_log is not None:
        _draw_log.append((B, N))
    if _fps_start_queue:
        s = _fps_start_queue.pop(0)
        return torch.as_tensor(s, dtype=torch.long).to(device)
    # reference: torch.randint(0, N, (B,), dtype=torch.long).to(device) -- CPU generator, then copied (:77)
    return torch.randint(0, N, (B,), dtype=torch.long).to(device, non_blocking=True)


def farthest_point_sample(xyz, npoint):
    """xyz [B,N,3] -> sampled indices i64 [B,npoint] (:65-86)."""
    B, N, _ = xyz.shape
    return ops.fps(xyz, npoint, _draw_fps_start(B, N, xyz.device))


def query_ball_point(radius, nsample, xyz, new_xyz):
    """-> i64 [B,S,nsample]: first `nsample` in-ball indices in index order, padded with the first (:89-109)."""
    return ops.ball_query(radius, nsample, xyz, new_xyz)


def sample_indices(npoint, radius, nsample, xyz):
    """The sampling half of sample_and_group: (new_xyz [B,npoint,3], idx i64 [B,npoint,nsample]) from a supplied plan or FPS + ball query."""
    B, N, _ = xyz.shape
    plan = _take_prefetched(xyz, npoint, radius, nsample) if _prefetched else None
    if plan is not None:
        _, new_xyz, idx = plan
    else:
        _, new_xyz = ops.fps(xyz, npoint, _draw_fps_start(B, N, xyz.device), return_xyz=True)
        idx = ops.ball_query(radius, nsample, xyz, new_xyz)
    return new_xyz, idx


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False, full_points=None, _pad_to=1,
                     _xyz_last=False):
    """FPS -> ball query -> gather/centre/concat (:112-148).  xyz [B,N,3], points [B,N,D] or None.
    Returns new_xyz [B,npoint,3], new_points [B,npoint,nsample,3+D] (xyz channels first).
    (_pad_to is internal: the set-abstraction modules ask for rows padded to a multiple of 4 floats.)"""
    B, N, _ = xyz.shape
    plan = _take_prefetched(xyz, npoint, radius, nsample) if _prefetched else None
    if plan is not None:
        fps_idx, new_xyz, idx = plan
    else:
        fps_idx, new_xyz = ops.fps(xyz, npoint, _draw_fps_start(B, N, xyz.device), return_xyz=True)
        idx = ops.ball_query(radius, nsample, xyz, new_xyz)
    if points is not None:
        new_points = ops.group(xyz, points, new_xyz, idx, xyz_last=_xyz_last, pad_to=_pad_to)
    elif full_points is not None:
        new_points = ops.index_points(full_points, idx)  # un-centred full features (:139-141)
    else:
        g = _grouped_xyz.get(idx.data_ptr()) if _grouped_xyz else None      # prepared with the sampling plan (harness)
        if g is not None and tuple(g.shape) == (B, npoint, nsample, (3 + _pad_to - 1) // _pad_to * _pad_to):
            new_points = g
        else:
            new_points = ops.group(xyz, None, new_xyz, idx, pad_to=_pad_to)
    if returnfps:
        return new_xyz, new_points, ops.index_points(xyz, idx), fps_idx
    return new_xyz, new_points


def sample_and_group_all(xyz, points):
    """One group holding every point, new_xyz = 0, coordinates NOT centred (:151-168)."""
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device)
    grouped = xyz.view(B, 1, N, C)
    if points is not None:
        grouped = torch.cat([grouped, points.reshape(B, 1, N, -1)], dim=-1)
    return new_xyz, grouped


def _points_major(t):
    """[B,C,N] -> [B,N,C]; free when `t` is the permuted view a previous layer of this module returned."""
    return t.permute(0, 2, 1).contiguous()


class PointNetSetAbstraction(nn.Module):
    """Same constructor, parameters and state_dict keys as the reference class (:171-216):
    mlp_convs.{i}.{weight[Co,Ci,1,1],bias}, mlp_bns.{i}.{weight,bias,running_mean,running_var,num_batches_tracked}.
    `mlp_dtype` ("f32" | "bf16", not a constructor argument: set it on the instance) selects the operand type of the grouped
    MLP's matrix-core contractions (sa_mlp.shared_mlp_max)."""
    mlp_dtype = "f32"
    sync_bn = None        # True / a process group: train-mode statistics over the global batch (sync_bn.enable)

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super().__init__()
        self.npoint = npoint
        self.radius = radius
        self.nsample = nsample
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last = out_channel
        self.group_all = group_all

    def forward(self, xyz, points, full_points=None):
        """xyz [B,3,N], points [B,D,N] or None -> new_xyz [B,3,S], new_points [B,D',S]."""
        xyz = _points_major(xyz)
        points = None if points is None else _points_major(points)
        full_points = None if full_points is None else _points_major(full_points)
        # internal channel order: features first, coordinates last, rows padded to a multiple of 4 floats (sa_mlp.py)
        # derived from the branch sample_and_group actually takes: `points` wins over `full_points` (:136-141)
        layout = "feats_first" if points is not None else "xyz_first"
        grad_to = None
        if self.group_all:
            B, N, C = xyz.shape
            new_xyz = _zeros((B, 1, C), xyz.device)
            if points is None:
                grouped = xyz.view(B, 1, N, C)
            else:
                pad = (-(C + points.shape[2])) % 4
                # (assembled outside autograd: the level's backward hands the gradient of `points` back compact -- no slice of the
                # concatenated rows' gradient, no copy to make it contiguous, no clear of the coordinate columns)
                grad_to = points if (points.requires_grad and torch.is_grad_enabled() and points.shape[2] % 4 == 0) else None
                parts = [points.detach() if grad_to is not None else points, xyz] + ([_zeros((B, N, pad), xyz.device)] if pad else [])
                grouped = torch.cat(parts, dim=-1).view(B, 1, N, -1)
        elif (points is not None and full_points is None and sa_mlp.FACTORED_FIRST in ("1", True)
              and sa_mlp.factored_supported(points, self.nsample, self.mlp_convs, self.mlp_bns, self.mlp_dtype, self.sync_bn)):
            # the first layer factorised: a linear map per source point, then a gather-add (sa_mlp.shared_mlp_max_factored)
            new_xyz, idx = sample_indices(self.npoint, self.radius, self.nsample, xyz)
            new_points = sa_mlp.shared_mlp_max_factored(xyz, points, new_xyz, idx, self.mlp_convs, self.mlp_bns, "xyz_first", dtype=self.mlp_dtype,
                                                        sync_bn=self.sync_bn)
            return new_xyz.permute(0, 2, 1), new_points.permute(0, 2, 1)
        else:
            new_xyz, grouped = sample_and_group(self.npoint, self.radius, self.nsample, xyz, points,
                                                full_points=full_points, _pad_to=4, _xyz_last=True)
        new_points = sa_mlp.shared_mlp_max(grouped, self.mlp_convs, self.mlp_bns, layout=layout, dtype=self.mlp_dtype, sync_bn=self.sync_bn,
                                           grad_to=grad_to)  # [B,S,C']
        return new_xyz.permute(0, 2, 1), new_points.permute(0, 2, 1)


class PointNetSetAbstractionMsg(nn.Module):
    """Multi-scale grouping (:219-276): one FPS, per-radius ball query + MLP, outputs concatenated over scales.
    Channel order inside a group is FEATURES first, centred xyz last (:262).  state_dict keys conv_blocks.{i}.{j}.*,
    bn_blocks.{i}.{j}.* as in the reference.  `mlp_dtype` as for PointNetSetAbstraction."""
    mlp_dtype = "f32"
    sync_bn = None

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list):
        super().__init__()
        self.npoint = npoint
        self.radius_list = radius_list
        self.nsample_list = nsample_list
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        for mlp in mlp_list:
            convs, bns = nn.ModuleList(), nn.ModuleList()
            last = in_channel + 3
            for out_channel in mlp:
                convs.append(nn.Conv2d(last, out_channel, 1))
                bns.append(nn.BatchNorm2d(out_channel))
                last = out_channel
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)

    def forward(self, xyz, points):
        xyz = _points_major(xyz)
        points = None if points is None else _points_major(points)
        B, N, _ = xyz.shape
        plan = _take_prefetched(xyz, self.npoint, self.radius_list, self.nsample_list) if _prefetched else None
        if plan is not None:        # (fps_idx, new_xyz, (idx per radius)): sampled ahead on a side stream (harness.TrainStep)
            _, new_xyz, idxs = plan
        else:
            _, new_xyz = ops.fps(xyz, self.npoint, _draw_fps_start(B, N, xyz.device), return_xyz=True)
            # (:255-258 loops query_ball_point over the radii on the same query / cloud pair: one scan, one distance per pair)
            idxs = ops.ball_query_multi(self.radius_list, self.nsample_list, xyz, new_xyz)
        outs = []
        for si, (radius, K, convs, bns) in enumerate(zip(self.radius_list, self.nsample_list, self.conv_blocks, self.bn_blocks)):
            idx = idxs[si] if idxs is not None else ops.ball_query(radius, K, xyz, new_xyz)
            if (points is not None and sa_mlp.FACTORED_FIRST in ("1", "msg", True)
                    and sa_mlp.factored_supported(points, K, convs, bns, self.mlp_dtype, self.sync_bn)):
                # first layer factorised (features: a linear map per source point; then a gather-add): no grouped tensor
                outs.append(sa_mlp.shared_mlp_max_factored(xyz, points, new_xyz, idx, convs, bns, "xyz_last", dtype=self.mlp_dtype, sync_bn=self.sync_bn))
                continue
            grouped = _grouped_xyz.get(idx.data_ptr()) if (points is None and _grouped_xyz) else None     # prepared with the sampling plan (harness)
            if grouped is None or tuple(grouped.shape) != (B, self.npoint, K, 4):
                grouped = ops.group(xyz, points, new_xyz, idx, xyz_last=True, pad_to=4)
            outs.append(sa_mlp.shared_mlp_max(grouped, convs, bns, dtype=self.mlp_dtype, sync_bn=self.sync_bn))
        return new_xyz.permute(0, 2, 1), torch.cat(outs, dim=-1).permute(0, 2, 1)


class PointNetFeaturePropagation(nn.Module):
    """Feature propagation (:279-329): 3-NN inverse-distance interpolation of the coarse features onto the dense points,
    concatenation with the dense skip features, then Conv1d + BatchNorm1d + ReLU layers.  Same constructor, parameters
    and state_dict keys as the reference (mlp_convs.{i}.{weight[Co,Ci,1],bias}, mlp_bns.{i}.*).  The interpolation is the
    HIP part (ops.three_nn / ops.three_interpolate); the per-point MLP is a plain GEMM chain and stays on rocBLAS.
    Gradients flow to points1 / points2 and the parameters, not to the coordinates."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last = out_channel

    def forward(self, xyz1, xyz2, points1, points2):
        """xyz1 [B,3,N], xyz2 [B,3,S], points1 [B,D1,N] or None, points2 [B,D2,S] -> [B,D',N]."""
        xyz1 = _points_major(xyz1)
        xyz2 = _points_major(xyz2)
        points2 = _points_major(points2)
        B, N, _ = xyz1.shape
        S = xyz2.shape[1]
        if S == 1:
            interpolated = points2.repeat(1, N, 1)                      # :307-308
        else:
            idx, weight = ops.three_nn(xyz1, xyz2)
            interpolated = ops.three_interpolate(points2, idx, weight)
        x = interpolated if points1 is None else torch.cat([_points_major(points1), interpolated], dim=-1)
        x = x.reshape(B * N, -1)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):             # Conv1d(k=1) over [B,C,N] == Linear over rows
            x = torch.relu(bn(torch.nn.functional.linear(x, conv.weight.squeeze(-1), conv.bias)))
        return x.view(B, N, -1).permute(0, 2, 1)


def timeit(tag, t):
    """models/pointnet2_utils.py:9-11."""
    import time
    print("{}: {}s".format(tag, time.time() - t))
    return time.time()


def pc_normalize(pc):
    """models/pointnet2_utils.py:13-19 (numpy, host): centre and scale into the unit sphere."""
    import numpy as np
    pc = pc - np.mean(pc, axis=0)
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))

