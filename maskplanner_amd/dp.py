"""Batch data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference is single-process / single-GPU (train_maskplanner.py:150) and has no collective.  The hot path
shards by sample: FPS, ball query, grouping, kNN and the mask matching are per-cloud, so the only
exchange a training step needs is the gradient all-reduce.  MaskPlanner's gradient is 105-358 MB of fp32, >97 %
of it in three head matrices (fc3, fc_normals, sm_fc3) whose gradients are produced FIRST in backward: buckets
are laid out in reverse registration order so those big buckets are on the wire while the encoder backward runs.

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is bound by one link; buckets are
kept large (default 64 MB) to stay in the bandwidth regime and few enough that per-collective latency is noise.

Each bucket owns one flat buffer.  Autograd writes gradients as usual (no pre-existing .grad, so no accumulate pass);
when the last gradient of a bucket is ready its members are packed into the flat buffer with ONE multi-tensor copy,
`param.grad` is re-pointed at views of that buffer, and the all-reduce (average) is started on the flat buffer: no
unpack copy afterwards, the optimizer reads the reduced values in place.
"""
import os

import torch
import torch.distributed as dist

# Test hook: run the collectives even in a 1-rank process group.  A gpurun box has one GPU and RCCL refuses two ranks
# on one device, so this is the only way to execute the bucket / stream / RCCL code on real hardware before the
# multi-GPU bench does (tests/test_gpu_modules.py::test_rccl_single_rank_collectives_do_not_change_the_step).
FORCE_COLLECTIVES = bool(os.environ.get("MASKPLANNER_FORCE_COLLECTIVES"))


def exchanging(process_group=None):
    """True when gradients / factors have to go through a collective."""
    if not dist.is_initialized():
        return False
    return dist.get_world_size(process_group) > 1 or FORCE_COLLECTIVES


class BucketedGradAllReduce:
    def __init__(self, params, bucket_bytes=64 << 20, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [p for p in params if p.requires_grad]
        self.params = params
        self.buckets = []
        self._views = []
        self._handles, self._pending, self._hooks = [], [], []
        self.deferred = False   # True: no exchange from the gradient hooks, everything in finish()
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"  # gloo has no AVG
        self.active = exchanging(process_group)
        if not self.active:
            return  # nothing to exchange: let autograd write .grad directly (no flat buffers, no extra add/zero passes)
        order = list(reversed(params))  # heads first == the order gradients become ready in backward
        cur, cur_bytes = [], 0
        for p in order:
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._pending = [len(ps) for _, ps in self.buckets]
        self._done = [False] * len(self.buckets)
        if self.active:
            for bi, (_, ps) in enumerate(self.buckets):
                for p in ps:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _close(self, ps):
        flat = torch.zeros(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
        views, off = [], 0
        for p in ps:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.buckets.append((flat, ps))
        self._views.append(views)

    def _reduce_bucket(self, bi):
        flat, ps = self.buckets[bi]
        views = self._views[bi]
        have = [(v, p.grad) for v, p in zip(views, ps) if p.grad is not None]
        if len(have) != len(ps):
            flat.zero_()  # a parameter without gradient this step contributes zeros
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])  # one fused pack kernel per bucket
        for p, v in zip(ps, views):
            p.grad = v
        # RCCL runs on its own stream and is ordered after everything already queued on the compute stream
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self._handles.append(dist.all_reduce(flat, op=op, group=self.group, async_op=True))
        self._done[bi] = True

    def _make_hook(self, bi):
        def hook(_param):
            if self.deferred:        # all buckets are exchanged in finish() (steps replayed from a graph fire no hooks at all)
                return
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._reduce_bucket(bi)
        return hook

    def zero_grad(self):
        """Replaces optimizer.zero_grad(): keeps the grad views alive."""
        if not self.active:
            for p in self.params:
                p.grad = None
            return
        for p in self.params:
            p.grad = None  # autograd then WRITES the next gradients instead of accumulating into the flat views
        self._pending = [len(ps) for _, ps in self.buckets]
        self._done = [False] * len(self.buckets)

    def rearm(self, grads):
        """For a backward pass that ran WITHOUT Python (a replayed graph): `grads` = [(param, its static gradient tensor)];
        point .grad back at them and mark every bucket as not yet exchanged, so that finish() packs and reduces them all."""
        for p, g in grads:
            p.grad = g
        if self.active:
            self._pending = [len(ps) for _, ps in self.buckets]
            self._done = [False] * len(self.buckets)

    def finish(self):
        """Wait for the in-flight all-reduces (the compute stream waits, not the host) and average."""
        if not self.active:
            return
        # a parameter that received no gradient this step never fired its hook: reduce its bucket now
        for bi, done in enumerate(self._done):
            if not done:
                self._reduce_bucket(bi)
        for h in self._handles:
            h.wait()
        self._handles = []
        if not self._avg:
            for flat, _ in self.buckets:
                flat.div_(self.world)

    def grad_bytes(self):
        return sum(p.numel() * p.element_size() for p in self.params)


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # "nccl" is RCCL on ROCm.  MASKPLANNER_DIST_BACKEND=gloo lets several ranks share one GPU for a functional
            # dry run of the multi-rank control flow where RCCL (one GPU per rank) cannot run.
            backend = os.environ.get("MASKPLANNER_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world
