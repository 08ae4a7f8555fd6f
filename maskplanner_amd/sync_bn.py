"""SyncBN option of the data-parallel path: train-mode BatchNorm statistics over the GLOBAL batch.

The reference is single-process, so its BatchNorm layers (9 x BatchNorm2d in sa1..sa3, models/pointnet2_utils.py:208-213; 4 x
BatchNorm1d in the heads, models/pointnet2_cls_ssg.py:273-275, 292-293) always normalise over the whole batch.  Sharding the
batch over GPUs gives per-replica statistics unless they are exchanged; with this option on, a data-parallel run reproduces
the single-device global-batch run (tests/test_gpu_dp.py: 2 ranks vs 1 process at 1e-5).

  * set abstraction levels: the fused kernels exchange their fp64 sums through mp_sa_mlp_{fwd,bwd}_ex's `sync` hook -- one
    small all-reduce (2 * C doubles) per BatchNorm layer and pass, issued from the callback below;
  * heads: `bn_relu_rows_sync`, the same algebra for a [B, C] activation in torch ops (any device, any backend).

`enable(model, group)` switches every BatchNorm of a drop-in model over.  Collectives inside the step rule out hipGraph replay:
harness.TrainStep launches eagerly with SyncBN on.
"""
import torch
import torch.distributed as dist

from . import _lib


def _active(group):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


class Exchange:
    """The mp_syncbn_t of one fused-MLP call: an fp64 device buffer + the all-reduce callback the library invokes."""

    def __init__(self, group, cmax, device):
        self.group = group
        self.world = dist.get_world_size(group)
        self.buf = torch.empty(4 * int(cmax), dtype=torch.float64, device=device)
        self.error = None
        base = self.buf.data_ptr()

        def allreduce(_user, ptr, count, _stream):
            # the library has queued the sums on torch's current stream; torch.distributed orders the collective behind it
            try:
                off = (int(ptr) - base) // 8
                dist.all_reduce(self.buf[off:off + int(count)], group=self.group)
                return 0
            except Exception as exc:        # never let an exception cross the C frame
                self.error = exc
                return 1

        self._cb = _lib.ALLREDUCE_FN(allreduce)
        self.struct = _lib.SyncBN(self._cb, None, self.world, self.buf.data_ptr())


def resolve(sync):
    """A module's `sync_bn` attribute -> process group to exchange over, or False when nothing is to be exchanged."""
    if sync is None or sync is False:
        return False
    group = None if sync is True else sync
    return group if _active(group) else False


class _BnReluRowsSync(torch.autograd.Function):
    """relu(batch_norm(x)) for x [B, C] with statistics over the global batch (all ranks hold equally many rows)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, group):
        world = dist.get_world_size(group)
        n = x.shape[0] * world
        xd = x.double()
        sums = torch.stack([xd.sum(0), (xd * xd).sum(0)])
        dist.all_reduce(sums, group=group)
        mean = sums[0] / n
        var = (sums[1] / n - mean * mean).clamp_min(0.0)
        rstd = torch.rsqrt(var + eps)
        if running_mean is not None:
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_((momentum * mean).to(running_mean.dtype))
                running_var.mul_(1 - momentum).add_((momentum * var * (n / max(n - 1, 1))).to(running_var.dtype))
        xhat = ((xd - mean) * rstd).float()
        y = torch.relu(xhat * gamma + beta)
        ctx.save_for_backward(xhat, y, gamma, rstd.float())
        ctx.group, ctx.n = group, n
        return y

    @staticmethod
    def backward(ctx, gy):
        xhat, y, gamma, rstd = ctx.saved_tensors
        dy = torch.where(y > 0, gy, torch.zeros_like(gy))
        local = torch.stack([dy.double().sum(0), (dy * xhat).double().sum(0)])
        glob = local.clone()
        dist.all_reduce(glob, group=ctx.group)
        dbeta, dgamma = local[0].float(), local[1].float()         # this rank's share; the gradient exchange averages them
        m1, m2 = (glob[0] / ctx.n).float(), (glob[1] / ctx.n).float()
        gx = (gamma * rstd) * (dy - m1 - xhat * m2)
        return gx, dgamma, dbeta, None, None, None, None, None


def bn_relu_rows_sync(x, bn, group):
    """F.relu(bn(x)) of a head block with global-batch statistics; running stats updated like nn.BatchNorm1d does (the caller
    advances num_batches_tracked)."""
    track = bn.track_running_stats and bn.running_mean is not None
    momentum = bn.momentum if bn.momentum is not None else 1.0 / max(float(bn.num_batches_tracked), 1.0)
    return _BnReluRowsSync.apply(x, bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None,
                                 momentum, bn.eps, group)


def enable(model, group=None, on=True):
    """Switch every BatchNorm of a drop-in model (set-abstraction levels and BatchNorm1d heads) to global-batch statistics over
    `group` (None = the default group).  Returns the model."""
    from .pointnet2_utils import PointNetSetAbstraction, PointNetSetAbstractionMsg
    value = (True if group is None else group) if on else None
    for m in model.modules():
        if isinstance(m, (PointNetSetAbstraction, PointNetSetAbstractionMsg, torch.nn.BatchNorm1d)):
            m.sync_bn = value
    return model
