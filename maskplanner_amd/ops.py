"""torch-facing wrappers of the C ABI (include/maskplanner_hip.h).

Every function takes HIP tensors, enqueues one or two kernels on torch's CURRENT stream and returns tensors that
live on the same device; nothing here synchronises with the host.  PyTorch supplies memory, streams and autograd
bookkeeping only -- all arithmetic on the hot path happens inside libmaskplanner_hip.so.  There is no CPU path:
a non-HIP tensor raises.
"""
import ctypes
import weakref

import torch

from . import _lib

MATCH_TOO_MANY_IDS, MATCH_PADDING_ID, MATCH_INFEASIBLE = 1, 2, 4     # include/maskplanner_hip.h MP_MATCH_*

# Scatter-add backwards use float atomics by default; set True for the fixed-order (bitwise reproducible) kernels.
DETERMINISTIC = False


def _need_hip(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("maskplanner_amd ops run on the MI355X only (HIP tensors); there is no CPU fallback")


def _f32(t):
    return t.contiguous() if t.dtype == torch.float32 else t.contiguous().float()


def _i64(t):
    return t.contiguous() if t.dtype == torch.int64 else t.contiguous().long()


def _p(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


class KernelTimer:
    """Optional per-entry-point device timing with HIP events recorded on the launch stream (bench.py's roofline
    leg).  `with KernelTimer() as kt: ...; kt.summary()` -> {op: (calls, mean_ms)}.  Events are only read in
    summary(), after the caller has synchronised: recording them does not stall the host."""
    active = None

    def __init__(self):
        self.events = {}

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def summary(self):
        out = {}
        for name, pairs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[name] = (len(ms), sum(ms) / max(len(ms), 1))
        return out


class ZeroArena:
    """One buffer for the small zero-initialised outputs of a training step (scatter targets of the nearest-neighbour backward, the
    wide heads' input gradients, dA of the factorised first layer): arm() clears it with ONE launch (mp_zero_arena_arm) and the library
    then skips its own clear of every output handed out by empty() -- five clear launches fewer per step.  Sizes are learnt: requests that
    do not fit fall back to an ordinary allocation (cleared by the library as before) and grow the arena for the next arm()."""
    # [r5] armed arenas by (device index, raw stream): two training steps on two streams of one process each have their own (the library's
    # table is keyed by the stream as well: csrc/api.hip)
    _armed = {}

    def __init__(self, device):
        self.device, self.buf, self.cap, self.cur, self.want = device, None, 0, 0, 0
        self._key, self._recorded, self._retired = None, False, []

    @staticmethod
    def current(device):
        """The arena armed on `device`'s current stream, or None."""
        if not ZeroArena._armed:
            return None
        idx = device.index if device.index is not None else _cur_device()
        return ZeroArena._armed.get((idx, _raw_stream(idx)))

    def arm(self, ticks=None, fticks=None):
        """Clear the arena on the current stream and make it the source of empty() until disarm().  ticks / fticks: int64 / float32
        device scalars advanced by 1 in the same launch (mp_zero_arena_arm_ticks); returns False when they were NOT advanced (no arena
        yet, too many, other dtypes) and the caller has to."""
        need = max(self.want, self.cur)
        capturing = torch.cuda.is_current_stream_capturing()
        if need > self.cap and not capturing:
            # (a buffer that was armed under capture has its address baked into recorded graphs -- the clear launch and every output carved
            # out of it: it is kept alive, never handed back to the allocator [ADVICE r4])
            if self.buf is not None and self._recorded:
                self._retired.append(self.buf)
            self.cap = (need + 4095) // 4096 * 4096
            self.buf = torch.empty((self.cap,), dtype=torch.uint8, device=self.device)
            self._recorded = False
        self._recorded = self._recorded or capturing
        self.cur, self.want = 0, 0
        idx = self.device.index if self.device.index is not None else _cur_device()
        key = (idx, _raw_stream(idx))
        other = ZeroArena._armed.get(key)
        if other is not None and other is not self:
            raise RuntimeError("another ZeroArena is armed on this stream (one owner per stream: disarm() it first)")
        if self._key is not None and self._key != key:
            ZeroArena._armed.pop(self._key, None)
        self._key = key
        ZeroArena._armed[key] = self
        ticks, fticks = list(ticks or []), list(fticks or [])
        if not self.cap:
            return not (ticks or fticks)
        if ((ticks or fticks) and len(ticks) <= 40 and len(fticks) <= 8
                and all(t.dtype == torch.int64 and t.numel() == 1 and t.device == self.buf.device for t in ticks)
                and all(t.dtype == torch.float32 and t.numel() == 1 and t.device == self.buf.device for t in fticks)):
            a64 = (ctypes.c_void_p * max(len(ticks), 1))(*[t.data_ptr() for t in ticks])
            a32 = (ctypes.c_void_p * max(len(fticks), 1))(*[t.data_ptr() for t in fticks])
            _run("zero_arena", self.buf, _lib.load().mp_zero_arena_arm_ticks, self.buf.data_ptr(), self.cap, len(ticks), a64, len(fticks), a32)
            return True
        _run("zero_arena", self.buf, _lib.load().mp_zero_arena_arm, self.buf.data_ptr(), self.cap)
        return not (ticks or fticks)

    def disarm(self):
        if self._key is not None and ZeroArena._armed.get(self._key) is self:
            del ZeroArena._armed[self._key]
            _lib.load().mp_zero_arena_disarm_stream(ctypes.c_void_p(self._key[1]))
        self._key = None

    def take(self, shape, dtype, device):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = (n * torch.empty((), dtype=dtype).element_size() + 255) // 256 * 256
        self.want += nbytes
        if self.buf is None or device != self.buf.device or self.cur + nbytes > self.cap:
            return None
        t = self.buf[self.cur:self.cur + nbytes].view(dtype)[:n].view(shape)
        self.cur += nbytes
        return t


def zeroed_empty(shape, dtype, device):
    """An output the LIBRARY zero-initialises before accumulating into it: from the armed ZeroArena when there is one (already clear),
    else an ordinary uninitialised allocation (the library clears it)."""
    a = ZeroArena.current(device)
    if a is not None:
        t = a.take(tuple(shape), dtype, device)
        if t is not None:
            return t
    return torch.empty(tuple(shape), dtype=dtype, device=device)


_raw_stream = torch._C._cuda_getCurrentRawStream     # hipStream_t of torch's current stream on a device index (no object churn)
_cur_device = torch._C._cuda_getDevice


def _run(op, ref, fn, *args):
    """Enqueue one C-ABI call on torch's current stream of ref's device and map its return code.  The common case (ref
    already on the current device, no timer) costs one ctypes call: a training step makes ~40 of these and the host is
    only ~15 % ahead of the GPU, so the torch.cuda.device / current_stream wrappers were measurable."""
    idx = ref.device.index
    if KernelTimer.active is None and idx == _cur_device():
        rc = fn(*args, _raw_stream(idx))
        if rc:
            _lib.check(rc, op)
        return
    kt = KernelTimer.active
    with torch.cuda.device(ref.device):
        if kt is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        rc = fn(*args, _stream(ref))
        if kt is not None:
            b.record()
            kt.events.setdefault(op, []).append((a, b))
    _lib.check(rc, op)


# ----------------------------------------------------------------------------------------------------------------
# index-producing ops (no autograd)
# ----------------------------------------------------------------------------------------------------------------
def _check_out(t, shape, dtype, like):
    if tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != like.device or not t.is_contiguous():
        raise ValueError(f"out tensor must be a contiguous {dtype} tensor of shape {tuple(shape)} on {like.device}")


@torch.no_grad()
def fps(xyz, npoint, start_idx, return_xyz=False, out=None):
    """farthest_point_sample with an explicit start (models/pointnet2_utils.py:65-86).  xyz [B,N,3].
    out=(idx i64 [B,npoint], new_xyz f32 [B,npoint,3] or None): write into existing contiguous tensors."""
    _need_hip(xyz, start_idx)
    xyz = _f32(xyz)
    if xyz.ndim != 3 or xyz.shape[2] != 3:
        raise ValueError("xyz must be [B,N,3]")
    B, N, _ = xyz.shape
    start_idx = _i64(start_idx)
    if out is not None:
        idx, new_xyz = out
        _check_out(idx, (B, npoint), torch.int64, xyz)
        if new_xyz is not None:
            _check_out(new_xyz, (B, npoint, 3), torch.float32, xyz)
        return_xyz = new_xyz is not None
    else:
        idx = torch.empty((B, npoint), dtype=torch.int64, device=xyz.device)
        new_xyz = torch.empty((B, npoint, 3), dtype=torch.float32, device=xyz.device) if return_xyz else None
    _run("fps", xyz, _lib.load().mp_fps_f32, _p(xyz), B, N, npoint, _p(start_idx), _p(idx), _p(new_xyz))
    return (idx, new_xyz) if return_xyz else idx


@torch.no_grad()
def ball_query(radius, nsample, xyz, new_xyz, out=None):
    """query_ball_point (models/pointnet2_utils.py:89-109).  xyz [B,N,3], new_xyz [B,S,3] -> i64 [B,S,nsample]."""
    _need_hip(xyz, new_xyz)
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    if out is not None:
        idx = out
        _check_out(idx, (B, S, nsample), torch.int64, xyz)
    else:
        idx = torch.empty((B, S, nsample), dtype=torch.int64, device=xyz.device)
    _run("ball_query", xyz, _lib.load().mp_ball_query_f32, _p(xyz), _p(new_xyz), B, N, S, float(radius), nsample, _p(idx))
    return idx


def ball_query_multi(radii, nsamples, xyz, new_xyz, out=None):
    """The radius loop of PointNetSetAbstractionMsg (models/pointnet2_utils.py:255-258: `query_ball_point(radius, K, xyz, new_xyz)` per scale)
    as ONE scan of the cloud (csrc/ball_query.hip: ball_query_multi_kernel): [i64 [B,S,K_r] per radius], each equal to ball_query(r, K_r)."""
    _need_hip(xyz, new_xyz)
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    n = len(radii)
    if n != len(nsamples) or n < 1:
        raise ValueError("ball_query_multi: one group size per radius")
    if n > 3:       # (the kernel carries three lists per query)
        return [ball_query(r, k, xyz, new_xyz, out=None if out is None else out[i]) for i, (r, k) in enumerate(zip(radii, nsamples))]
    idxs = []
    for i, k in enumerate(nsamples):
        if out is not None:
            _check_out(out[i], (B, S, k), torch.int64, xyz)
            idxs.append(out[i])
        else:
            idxs.append(torch.empty((B, S, k), dtype=torch.int64, device=xyz.device))
    import ctypes
    rr = (ctypes.c_double * n)(*[float(r) for r in radii])
    kk = (ctypes.c_int64 * n)(*[int(k) for k in nsamples])
    oo = (ctypes.c_void_p * n)(*[t.data_ptr() for t in idxs])
    with torch.cuda.device(xyz.device):
        rc = _lib.load().mp_ball_query_multi_f32(_p(xyz), _p(new_xyz), B, N, S, n, rr, kk, oo, _stream(xyz))
    if rc == _lib.MP_EUNSUPPORTED and n > 1 and N <= 13312 and max(nsamples) <= 1024:
        # the lists of all radii do not fit beside the cloud in LDS (large N x large group sizes): one scan per radius, as the reference does
        return [ball_query(r, k, xyz, new_xyz, out=idxs[i]) for i, (r, k) in enumerate(zip(radii, nsamples))]
    _lib.check(rc, "ball_query_multi")
    return idxs


@torch.no_grad()
def square_distance(src, dst):
    """square_distance (models/pointnet2_utils.py:21-42), expanded form.  src [B,S,3], dst [B,N,3] -> [B,S,N]."""
    _need_hip(src, dst)
    src, dst = _f32(src), _f32(dst)
    if src.shape[-1] != 3 or dst.shape[-1] != 3:
        raise ValueError("square_distance kernel is specialised for 3-D points")
    B, S, _ = src.shape
    N = dst.shape[1]
    out = torch.empty((B, S, N), dtype=torch.float32, device=src.device)
    _run("square_distance", src, _lib.load().mp_square_distance_f32, _p(src), _p(dst), B, S, N, _p(out))
    return out


# ---- per-batch quantities of the loss's TARGETS, off the step ----------------------------------------------------------------------
# The padded lengths of a ground-truth tensor (pytorch3d_chamfer.py:138-149) and the screening planes of the nearest-neighbour search
# that uses it as the reference set depend on the batch alone.  A training harness whose batch tensors are static device buffers
# registers them here (once for a resident batch; per batch, on the side stream, for streamed ones: refresh_static_target):
# padded_lengths() and the chamfer terms then pick the prepared tensors up by the target's address instead of launching.
_STATIC_TARGETS = {}


@torch.no_grad()
def register_static_target(y, planes=False, storage=None):
    """y [B, P2, D] float32 contiguous, a buffer whose CONTENT only changes together with its entry (compute_target_aux into the
    entry's tensors, or into a second set that is copied over them when y is rewritten); planes: also the K = 1 search's reference
    planes (y as the reference set).  storage: (lengths i64 [B], workspace u8 [target_aux_bytes(...)] or None) owned by the caller (a
    harness keeps them inside its double-buffered plan); without it the entry allocates its own and is filled here.
    Returns the entry {"lengths", "ws", "nws"}."""
    _need_hip(y)
    if y.dtype != torch.float32 or not y.is_contiguous() or y.ndim != 3:
        raise ValueError("static targets are contiguous float32 [B, P, D] tensors")
    B, P2, D = y.shape
    nws = target_aux_bytes(B, P2, D) if planes else 0
    if storage is None:
        lengths = torch.empty((B,), dtype=torch.int64, device=y.device)
        ws = torch.empty((nws,), dtype=torch.uint8, device=y.device) if nws else None
    else:
        lengths, ws = storage
    # (a weak reference: the entry dies with the tensor -- a later tensor at the same address must not inherit it)
    e = {"shape": tuple(y.shape), "lengths": lengths, "ws": ws if nws else None, "nws": nws, "ref": weakref.ref(y)}
    for k in [k for k, v in _STATIC_TARGETS.items() if v["ref"]() is None]:
        del _STATIC_TARGETS[k]
    _STATIC_TARGETS[y.data_ptr()] = e
    if storage is None:
        compute_target_aux(y, lengths, e["ws"])
    return e


def target_aux_bytes(B, P2, D):
    """Workspace bytes of a target's reference planes (0: the screened search does not take this shape)."""
    return int(_lib.load().mp_knn1_workspace_bytes(B, P2, D))


@torch.no_grad()
def compute_target_aux(src, lengths, ws):
    """Padded lengths of src [B, P2, D] into `lengths`, and (ws not None) the reference planes of src under those lengths into `ws`, on the
    current stream."""
    B, P2, D = src.shape
    lib = _lib.load()
    _run("padded_lengths", src, lib.mp_padded_lengths_f32, _p(src), B, P2, D, _p(lengths))
    if ws is not None:
        _run("knn1_prepare", src, lib.mp_knn1_prepare_f32, _p(src), _p(lengths), B, P2, D, _p(ws), ws.numel())


@torch.no_grad()
def refresh_static_target(y):
    """Recompute the registered entry of `y` in place on the current stream (after y was rewritten; nothing may be reading the entry)."""
    e = _STATIC_TARGETS[y.data_ptr()]
    compute_target_aux(y, e["lengths"], e["ws"])


def forget_static_targets():
    _STATIC_TARGETS.clear()


def _static_target(y):
    e = _STATIC_TARGETS.get(y.data_ptr())
    if e is None or e["shape"] != tuple(y.shape) or y.dtype != torch.float32:
        return None
    if e["ref"]() is None:
        del _STATIC_TARGETS[y.data_ptr()]
        return None
    return e


@torch.no_grad()
def padded_lengths(y):
    """pytorch3d_chamfer.py:138-149: first column with y[b,c,0] == -100, else P2.  -> i64 [B] on device."""
    _need_hip(y)
    e = _static_target(y)
    if e is not None:
        return e["lengths"]
    y = _f32(y)
    B, P2, D = y.shape
    out = torch.empty((B,), dtype=torch.int64, device=y.device)
    _run("padded_lengths", y, _lib.load().mp_padded_lengths_f32, _p(y), B, P2, D, _p(out))
    return out


@torch.no_grad()
def mask_match(pred_masks, target_ids, return_cost=False, target_value=None):
    """loss_handler.py:838-875 on device.  pred_masks [B,M,S] logits, target_ids [B,S] f32.
    Returns match_col i64 [B,M] (-1 = unmatched), uniq_ids f32 [B,64], n_targets i64 [B], status i32 [B]
    (0, or bits MATCH_TOO_MANY_IDS / MATCH_PADDING_ID / MATCH_INFEASIBLE: what the reference asserts on or scipy raises for;
    and the fp32 cost [B,M,64] when asked).  target_value [B,S]: smooth targets, MSE cost (:830, :959-964)."""
    _need_hip(pred_masks, target_ids)
    pred_masks, target_ids = _f32(pred_masks), _f32(target_ids)
    B, M, S = pred_masks.shape
    if target_value is not None:
        _need_hip(target_value)
        target_value = _f32(target_value)
        if tuple(target_value.shape) != (B, S):
            raise ValueError("target_value must be [B,S]")
    dev = pred_masks.device
    match = torch.empty((B, M), dtype=torch.int64, device=dev)
    uniq = torch.empty((B, _lib.MASK_CAP), dtype=torch.float32, device=dev)
    nt = torch.empty((B,), dtype=torch.int64, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    cost = torch.empty((B, M, _lib.MASK_CAP), dtype=torch.float32, device=dev) if return_cost else None
    _run("mask_match", pred_masks, _lib.load().mp_mask_match_f32, _p(pred_masks), _p(target_ids), _p(target_value), B, M, S, _p(match),
         _p(uniq), _p(nt), _p(cost), _p(status))
    return (match, uniq, nt, status, cost) if return_cost else (match, uniq, nt, status)


@torch.no_grad()
def lsap(costs):
    """scipy.optimize.linear_sum_assignment for a list of 2-D fp32 device cost matrices (hungarianMatcher.py:58-61), solved
    side by side on the GPU with scipy's algorithm and tie-breaking.  Returns (pairs, status): one (row_ind, col_ind) pair
    of i64 DEVICE tensors per matrix, rows ascending, len == min(shape) -- scipy's convention -- and the per-sample
    status tensor (0 = solved; non-zero = infeasible cost matrix, indices are -1)."""
    if not costs:
        return [], None
    _need_hip(*costs)
    dev = costs[0].device
    B = len(costs)
    tr = [c.shape[0] > c.shape[1] for c in costs]                 # scipy solves the transpose when rows > columns
    mats = [(_f32(c).t() if t else _f32(c)) for c, t in zip(costs, tr)]
    Rmax = max(m.shape[0] for m in mats)
    Cmax = max(m.shape[1] for m in mats)
    if Cmax > 2048:
        raise _lib.MaskPlannerHipError("lsap: more than 2048 columns per sample is outside the gfx950 kernel's range")
    pad = torch.zeros((B, max(Rmax, 1), max(Cmax, 1)), dtype=torch.float32, device=dev)
    for b, m in enumerate(mats):
        pad[b, :m.shape[0], :m.shape[1]] = m
    nr = torch.tensor([m.shape[0] for m in mats], dtype=torch.int32).to(dev)
    nc = torch.tensor([m.shape[1] for m in mats], dtype=torch.int32).to(dev)
    c4r = torch.empty((B, pad.shape[1]), dtype=torch.int64, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    _run("lsap", pad, _lib.load().mp_lsap_f32, _p(pad), B, pad.shape[1], pad.shape[2], pad.shape[2], pad.shape[1] * pad.shape[2],
         _p(nr), _p(nc), _p(c4r), _p(status))
    out = []
    for b, (m, t) in enumerate(zip(mats, tr)):
        n = m.shape[0]
        cols = c4r[b, :n]
        rows = torch.arange(n, dtype=torch.int64, device=dev)
        if t:   # solved on the transpose: (row, col) = (col4row[k], k), reported in ascending row order
            rows, order = torch.sort(cols)
            cols = order
        out.append((rows, cols))
    return out, status


@torch.no_grad()
def match_segments(outputs, targets):
    """models/hungarianMatcher.py:44-61 on the device: Euclidean cost of every sample's [S, Sgt_b] block in ONE launch
    (mp_cdist_batch_f32, straight into the solver's padded layout) + the batched LAP.  outputs [B,S,D]; targets: list of B
    tensors [Sgt_b, D].  Returns (pairs, status) like lsap()."""
    _need_hip(outputs, *targets)
    outputs = _f32(outputs)
    B, S, D = outputs.shape
    dev = outputs.device
    sizes = [int(t.shape[0]) for t in targets]
    flat = _f32(torch.cat([t.reshape(-1, D) for t in targets], dim=0)) if sum(sizes) else torch.zeros((0, D), device=dev)
    offsets = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0).tolist()), dtype=torch.int64).to(dev)
    Rmax = max(max(min(S, t) for t in sizes), 1)
    Cmax = max(max(max(S, t) for t in sizes), 1)
    if Cmax > 2048:
        raise _lib.MaskPlannerHipError("lsap: more than 2048 columns per sample is outside the gfx950 kernel's range")
    cost = torch.empty((B, Rmax, Cmax), dtype=torch.float32, device=dev)
    nr = torch.empty((B,), dtype=torch.int32, device=dev)
    nc = torch.empty((B,), dtype=torch.int32, device=dev)
    lib = _lib.load()
    _run("cdist_batch", outputs, lib.mp_cdist_batch_f32, _p(outputs), _p(flat) if flat.numel() else None, _p(offsets), B, S, D, Rmax, Cmax,
         _p(cost), _p(nr), _p(nc))
    c4r = torch.empty((B, Rmax), dtype=torch.int64, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    _run("lsap", cost, lib.mp_lsap_f32, _p(cost), B, Rmax, Cmax, Cmax, Rmax * Cmax, _p(nr), _p(nc), _p(c4r), _p(status))
    out = []
    for b, t in enumerate(sizes):
        n = min(S, t)
        cols = c4r[b, :n]
        rows = torch.arange(n, dtype=torch.int64, device=dev)
        if S > t:   # solved on the transpose: (row, col) = (col4row[k], k), reported in ascending row order
            rows, order = torch.sort(cols)
            cols = order
        out.append((rows, cols))
    return out, status


# ----------------------------------------------------------------------------------------------------------------
# differentiable ops
# ----------------------------------------------------------------------------------------------------------------
class _IndexPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        B, N, C = points.shape
        M = idx.numel() // B if B > 0 else 0
        out = torch.empty(tuple(idx.shape) + (C,), dtype=torch.float32, device=points.device)
        _run("index_points", points, _lib.load().mp_index_points_f32, _p(points), _p(idx), B, N, C, M, _p(out))
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, C, M)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        B, N, C, M = ctx.dims
        grad_out = _f32(grad_out)
        grad = torch.empty((B, N, C), dtype=torch.float32, device=grad_out.device)
        _run("index_points_bwd", grad_out, _lib.load().mp_index_points_bwd_f32, _p(grad_out), _p(idx), B, N, C, M, _p(grad), int(DETERMINISTIC))
        return grad, None


def index_points(points, idx):
    """index_points (models/pointnet2_utils.py:45-62): points [B,N,C], idx [B,...] -> [B,...,C]; grad w.r.t. points."""
    _need_hip(points, idx)
    if points.ndim != 3:
        raise ValueError("points must be [B,N,C]")
    return _IndexPoints.apply(_f32(points), _i64(idx))


@torch.no_grad()
def three_nn(xyz1, xyz2, return_dist=False):
    """models/pointnet2_utils.py:310-316: for every xyz1 [B,N,3] point its 3 nearest xyz2 [B,S,3] points (S >= 3) and the
    normalised inverse-distance weights.  -> (idx i64 [B,N,3], weight [B,N,3][, dist [B,N,3]]).  No autograd: point
    coordinates carry no gradient on this path."""
    _need_hip(xyz1, xyz2)
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    if xyz1.ndim != 3 or xyz2.ndim != 3 or xyz1.shape[2] != 3 or xyz2.shape[2] != 3 or xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("xyz1 must be [B,N,3] and xyz2 [B,S,3]")
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    if S < 3:
        raise ValueError("three_nn needs at least 3 source points")
    idx = torch.empty((B, N, 3), dtype=torch.int64, device=xyz1.device)
    w = torch.empty((B, N, 3), dtype=torch.float32, device=xyz1.device)
    dist = torch.empty((B, N, 3), dtype=torch.float32, device=xyz1.device) if return_dist else None
    _run("three_nn", xyz1, _lib.load().mp_three_nn_f32, _p(xyz1), _p(xyz2), B, N, S, _p(dist), _p(idx), _p(w))
    return (idx, w, dist) if return_dist else (idx, w)


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points2, idx, weight):
        B, S, D = points2.shape
        N = idx.shape[1]
        out = torch.empty((B, N, D), dtype=torch.float32, device=points2.device)
        _run("three_interpolate", points2, _lib.load().mp_three_interpolate_f32, _p(points2), _p(idx), _p(weight), B, N, S, D, _p(out))
        ctx.save_for_backward(idx, weight)
        ctx.dims = (B, N, S, D)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        B, N, S, D = ctx.dims
        grad_out = _f32(grad_out)
        grad = torch.empty((B, S, D), dtype=torch.float32, device=grad_out.device)
        _run("three_interpolate_bwd", grad_out, _lib.load().mp_three_interpolate_bwd_f32, _p(grad_out), _p(idx), _p(weight), B, N, S, D,
             _p(grad), int(DETERMINISTIC))
        return grad, None, None


def three_interpolate(points2, idx, weight):
    """models/pointnet2_utils.py:317: sum_k points2[b, idx[b,n,k], :] * weight[b,n,k]; grad w.r.t. points2."""
    _need_hip(points2, idx, weight)
    if points2.ndim != 3 or idx.ndim != 3 or idx.shape[2] != 3 or tuple(weight.shape) != tuple(idx.shape):
        raise ValueError("points2 must be [B,S,D], idx and weight [B,N,3]")
    return _ThreeInterpolate.apply(_f32(points2), _i64(idx), _f32(weight))


class _Group(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, feats, new_xyz, idx, xyz_last, stride):
        B, N, _ = xyz.shape
        _, S, K = idx.shape
        D = 0 if feats is None else feats.shape[2]
        out = torch.empty((B, S, K, stride), dtype=torch.float32, device=xyz.device)
        _run("group", xyz, _lib.load().mp_group_f32, _p(xyz), _p(feats), _p(new_xyz), _p(idx), B, N, S, K, D, int(xyz_last),
             stride, _p(out))
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, S, K, D, int(xyz_last), stride)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        B, N, S, K, D, xyz_last, stride = ctx.dims
        if D == 0 or not ctx.needs_input_grad[1]:
            return None, None, None, None, None, None
        grad_out = _f32(grad_out)
        grad = torch.empty((B, N, D), dtype=torch.float32, device=grad_out.device)
        _run("group_bwd", grad_out, _lib.load().mp_group_bwd_f32, _p(grad_out), _p(idx), B, N, S, K, D, xyz_last, stride,
             _p(grad), int(DETERMINISTIC))
        return None, grad, None, None, None, None


@torch.no_grad()
def group_xyz_into(xyz, new_xyz, idx, out):
    """ops.group(xyz, None, new_xyz, idx, pad_to=out.shape[-1]) into a caller's buffer [B, S, K, stride] (no autograd: coordinates)."""
    _need_hip(xyz, new_xyz, idx, out)
    B, N, _ = xyz.shape
    _, S, K = idx.shape
    _run("group", xyz, _lib.load().mp_group_f32, _p(xyz), None, _p(new_xyz), _p(idx), B, N, S, K, 0, 0, out.shape[-1], _p(out))
    return out


def group(xyz, feats, new_xyz, idx, xyz_last=False, pad_to=1):
    """sample_and_group tail (models/pointnet2_utils.py:133-143; MSG order :258-262 when xyz_last):
    [B,S,K,3+D] = cat(xyz[idx] - new_xyz, feats[idx]).  Differentiable w.r.t. feats (the coordinates are network
    inputs / FPS selections and carry no gradient on this path).  pad_to > 1 rounds the row length up to a multiple
    of it with zero columns (the fused MLP consumes rows of a multiple of 4 floats)."""
    _need_hip(xyz, feats, new_xyz, idx)
    if xyz.requires_grad or new_xyz.requires_grad:
        raise NotImplementedError("group(): gradients w.r.t. coordinates are not part of the hot path; "
                                  "compose index_points() for that")
    C = 3 + (0 if feats is None else feats.shape[2])
    stride = (C + pad_to - 1) // pad_to * pad_to
    return _Group.apply(_f32(xyz), None if feats is None else _f32(feats), _f32(new_xyz), _i64(idx), bool(xyz_last), stride)


def _knn_workspace(like, B, P2, D, K):
    """Device scratch of the screened K = 1 search (bf16 planes + norms of the references, mp_knn1_workspace_bytes); (None, 0) when the
    library would use the direct scan anyway."""
    if K != 1:
        return None, 0
    n = int(_lib.load().mp_knn1_workspace_bytes(B, P2, D))
    if n <= 0:
        return None, 0
    return torch.empty((n,), dtype=torch.uint8, device=like.device), n


class _Knn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p1, p2, len1, len2, K):
        B, P1, D = p1.shape
        P2 = p2.shape[1]
        dists = torch.empty((B, P1, K), dtype=torch.float32, device=p1.device)
        idx = torch.empty((B, P1, K), dtype=torch.int64, device=p1.device)
        ws, nws = _knn_workspace(p1, B, P2, D, K)
        _run("knn", p1, _lib.load().mp_knn_f32, _p(p1), _p(p2), _p(len1), _p(len2), B, P1, P2, D, K, _p(dists), _p(idx), _p(ws), nws)
        ctx.save_for_backward(p1, p2, len1, len2, idx)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)      # (a zero gradient for the int64 index output would be a fill launch per call)
        ctx.K = K
        return dists, idx

    @staticmethod
    def backward(ctx, grad_dists, _grad_idx):
        if grad_dists is None:
            return None, None, None, None, None
        p1, p2, len1, len2, idx = ctx.saved_tensors
        B, P1, D = p1.shape
        P2 = p2.shape[1]
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g1 = torch.empty_like(p1) if need1 else None
        g2 = (torch.empty_like(p2) if DETERMINISTIC else zeroed_empty(p2.shape, torch.float32, p2.device)) if need2 else None
        if need1 or need2:
            grad_dists = _f32(grad_dists)
            _run("knn_bwd", p1, _lib.load().mp_knn_bwd_f32, _p(p1), _p(p2), _p(len1), _p(len2), _p(idx), _p(grad_dists), B, P1, P2, D, ctx.K, _p(g1), _p(g2), int(DETERMINISTIC))
        return g1, g2, None, None, None


def knn(p1, p2, lengths1=None, lengths2=None, K=1):
    """pytorch3d.ops.knn.knn_points contract: (dists [B,P1,K] squared L2 ascending, idx [B,P1,K] i64)."""
    _need_hip(p1, p2, lengths1, lengths2)
    if p1.ndim != 3 or p2.ndim != 3 or p1.shape[0] != p2.shape[0] or p1.shape[2] != p2.shape[2]:
        raise ValueError("pts1 and pts2 must be [B,P,D] with equal batch and feature dimensions")
    l1 = None if lengths1 is None else _i64(lengths1)
    l2 = None if lengths2 is None else _i64(lengths2)
    return _Knn.apply(_f32(p1), _f32(p2), l1, l2, int(K))


_REDUCE_COUNTERS = {}


def _reduce_counter(device):
    """The zero-on-entry / zero-on-exit device counter of the one-launch reductions.  Launches on one stream run one after the other
    and may share a counter; launches on two streams (evaluation metrics next to the training stream) must not, so eager launches
    get one counter per (device, stream).  Launches recorded into a hipGraph get the device's "graph" counter: replays of the
    training step's graphs are ordered among themselves, and the counter is created by the first EAGER call (the harness runs
    eager steps before it records), so no allocation or fill lands inside a capture."""
    graph_key = (device, "graph")
    if graph_key not in _REDUCE_COUNTERS and not torch.cuda.is_current_stream_capturing():
        _REDUCE_COUNTERS[graph_key] = torch.zeros((1,), dtype=torch.int32, device=device)
    key = graph_key if torch.cuda.is_current_stream_capturing() else (device, torch.cuda.current_stream(device).cuda_stream)
    t = _REDUCE_COUNTERS.get(key)
    if t is None:
        t = torch.zeros((1,), dtype=torch.int32, device=device)
        _REDUCE_COUNTERS[key] = t
    return t


class GradAccum:
    """One gradient buffer for several chamfer terms on the same prediction (the asymmetric composite losses take three nearest-
    neighbour terms of `y_pred`, loss_handler.py:604-664): each term's backward adds its part into the buffer and only the LAST one to
    run hands it to autograd -- no fan-out add launches.  Needs the armed ZeroArena (the buffer starts zero and the library must not
    clear it between the terms); otherwise, and under ops.DETERMINISTIC, the terms fall back to their own gradients."""

    def __init__(self, base):
        self.ptr, self.numel, self.shape = base.data_ptr(), base.numel(), tuple(base.shape)
        self.pending, self.buf, self.off = 0, None, False

    def operand(self, p1, p2):
        """Which operand of a term is the shared prediction (1, 2) or 0."""
        if p1.data_ptr() == self.ptr and p1.numel() == self.numel:
            return 1
        if p2.data_ptr() == self.ptr and p2.numel() == self.numel:
            return 2
        return 0

    def buffer(self, device):
        if self.off:
            return None
        if self.buf is None:
            a = ZeroArena.current(device)
            self.buf = a.take(self.shape, torch.float32, device) if (a is not None and not DETERMINISTIC) else None
            if self.buf is None:
                self.off = True
        return self.buf


class _ChamferTerm(torch.autograd.Function):
    """One reduced, one-directional chamfer term (pytorch3d_chamfer.py:257-334 with asymmetric / reverse_asymmetric): nearest
    neighbour of every row of p1 in p2 (K = 1), sum or mean over the rows, sum or mean over the batch, times `scale`, plus the
    running total `add`.  Forward = the knn launch + the reduction; backward = ONE launch (mp_knn_bwd_reduced_f32: the per-row
    gradient scale / div / len is formed inside the scatter).  Returns (value, dists [B,P1], idx [B,P1]); the last two carry no
    gradient."""

    @staticmethod
    def forward(ctx, p1, p2, len1, len2, point_mean, batch_mode, div, scale, add, accum=None):
        ctx.accum, ctx.which = None, 0
        if accum is not None:
            ctx.which = accum.operand(p1, p2)
            # (a term counts only if its shared operand takes a gradient at all; every counted term MUST take part in the backward pass --
            # the last one to run hands the total over -- which holds for the composite losses that sum all their terms)
            if ctx.which and ctx.needs_input_grad[ctx.which - 1]:
                ctx.accum = accum
                accum.pending += 1
        B, P1, D = p1.shape
        P2 = p2.shape[1]
        lib = _lib.load()
        dists = torch.empty((B, P1), dtype=torch.float32, device=p1.device)
        idx = torch.empty((B, P1), dtype=torch.int64, device=p1.device)
        e = _static_target(p2)
        if e is not None and e["ws"] is not None and len2 is not None and len2.data_ptr() == e["lengths"].data_ptr():
            # the references are a registered target: their planes were prepared with the batch (register_static_target)
            _run("knn", p1, lib.mp_knn1_prepared_f32, _p(p1), _p(p2), _p(len1), _p(len2), B, P1, P2, D, _p(dists), _p(idx), _p(e["ws"]), e["nws"])
        else:
            ws, nws = _knn_workspace(p1, B, P2, D, 1)
            _run("knn", p1, lib.mp_knn_f32, _p(p1), _p(p2), _p(len1), _p(len2), B, P1, P2, D, 1, _p(dists), _p(idx), _p(ws), nws)
        out = torch.empty((B,) if batch_mode == 0 else (), dtype=torch.float32, device=p1.device)
        scratch = torch.empty((B,), dtype=torch.float32, device=p1.device) if batch_mode != 0 else None
        if batch_mode != 0:
            _run("chamfer_reduce", dists, lib.mp_chamfer_reduce1_f32, _p(dists), _p(len1), B, P1, int(point_mean), int(batch_mode),
                 float(div), float(scale), _p(scratch), _p(out), _p(add), _p(_reduce_counter(p1.device)))
        else:
            _run("chamfer_reduce", dists, lib.mp_chamfer_reduce_f32, _p(dists), _p(len1), B, P1, int(point_mean), int(batch_mode),
                 float(div), float(scale), _p(scratch), _p(out), _p(add))
        ctx.save_for_backward(p1, p2, len1, len2, idx)
        ctx.meta = (int(point_mean), int(batch_mode), float(div), float(scale))
        ctx.mark_non_differentiable(dists, idx)
        ctx.set_materialize_grads(False)
        return out, dists, idx

    @staticmethod
    def backward(ctx, grad_out, _gd, _gi):
        if grad_out is None:
            if ctx.accum is not None:       # [r5, ADVICE r4] a counted term leaves the count on EVERY exit
                ctx.accum.pending -= 1
                if ctx.accum.pending == 0 and ctx.accum.buf is not None:       # ... and the last one hands the total over even if it adds nothing
                    g = ctx.accum.buf
                    return ((g.view(ctx.saved_tensors[0].shape) if ctx.which == 1 else None), (g.view(ctx.saved_tensors[1].shape) if ctx.which == 2 else None)) + (None,) * 8
            return (None,) * 10
        p1, p2, len1, len2, idx = ctx.saved_tensors
        point_mean, batch_mode, div, scale = ctx.meta
        B, P1, D = p1.shape
        P2 = p2.shape[1]
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        flags = int(DETERMINISTIC)
        shared = ctx.accum.buffer(p1.device) if ctx.accum is not None else None
        if shared is not None and ctx.which == 1:
            g1, flags = shared.view(p1.shape), flags | 2          # += into the shared buffer
        else:
            g1 = torch.empty_like(p1) if need1 else None
        if shared is not None and ctx.which == 2:
            g2 = shared.view(p2.shape)                            # the scatter's atomics add; inside the arena the library does not clear
        else:
            g2 = (torch.empty_like(p2) if DETERMINISTIC else zeroed_empty(p2.shape, torch.float32, p2.device)) if need2 else None
        if need1 or need2:
            grad_out = _f32(grad_out)
            _run("knn_bwd", p1, _lib.load().mp_knn_bwd_reduced_f32, _p(p1), _p(p2), _p(len1), _p(len2), _p(idx), _p(grad_out), point_mean,
                 batch_mode, div, scale, B, P1, P2, D, _p(g1), _p(g2), flags)
        if ctx.accum is not None:
            ctx.accum.pending -= 1
            if shared is not None and ctx.accum.pending > 0:        # only the last term to run hands the total to autograd
                if ctx.which == 1:
                    g1 = None
                else:
                    g2 = None
        return g1, g2, None, None, None, None, None, None, (grad_out if ctx.needs_input_grad[8] else None), None


RELU_TAP = None     # test hook (tests/test_gpu_routing.py): a list that receives the ReLU mask [B, C] of every bn_relu_rows call
KNN_TAP = None      # test hook (tests/test_gpu_routing.py): a list that receives the nearest-neighbour indices of every chamfer_term call


def chamfer_term(p1, p2, lengths1, lengths2, point_reduction="mean", batch_reduction="mean", scale=1.0, add=None, grad_accum=None):
    """(value, dists [B,P1], idx [B,P1]) of one reduced one-directional chamfer term, see _ChamferTerm; lengths1 is required
    (the point mean divides by it).  Same numbers as knn(K=1) followed by chamfer_reduce."""
    _need_hip(p1, p2, lengths1, lengths2, add)
    if p1.ndim != 3 or p2.ndim != 3 or p1.shape[0] != p2.shape[0] or p1.shape[2] != p2.shape[2]:
        raise ValueError("pts1 and pts2 must be [B,P,D] with equal batch and feature dimensions")
    batch_mode = {None: 0, "sum": 1, "mean": 2}[batch_reduction]
    if add is not None and (batch_mode == 0 or add.numel() != 1 or add.dtype != torch.float32):
        raise ValueError("add must be a float32 scalar and needs a batch reduction")
    if point_reduction not in ("mean", "sum"):
        raise ValueError("point_reduction must be 'mean' or 'sum'")
    res = _ChamferTerm.apply(_f32(p1), _f32(p2), _i64(lengths1), None if lengths2 is None else _i64(lengths2),
                             point_reduction == "mean", batch_mode, float(p1.shape[0]), float(scale), add, grad_accum)
    if KNN_TAP is not None:
        KNN_TAP.append(res[2])
    return res


class _ChamferReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cham, lengths, point_mean, batch_mode, div, scale, add):
        N, P = cham.shape
        out = torch.empty((N,) if batch_mode == 0 else (), dtype=torch.float32, device=cham.device)
        scratch = torch.empty((N,), dtype=torch.float32, device=cham.device) if batch_mode != 0 else None
        _run("chamfer_reduce", cham, _lib.load().mp_chamfer_reduce_f32, _p(cham), _p(lengths), N, P, int(point_mean), int(batch_mode),
             float(div), float(scale), _p(scratch), _p(out), _p(add))
        ctx.save_for_backward(lengths)
        ctx.meta = (N, P, int(point_mean), int(batch_mode), float(div), float(scale))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (lengths,) = ctx.saved_tensors
        N, P, point_mean, batch_mode, div, scale = ctx.meta
        grad_out = _f32(grad_out)
        grad = torch.empty((N, P), dtype=torch.float32, device=grad_out.device)
        _run("chamfer_reduce_bwd", grad_out, _lib.load().mp_chamfer_reduce_bwd_f32, _p(grad_out), _p(lengths), N, P, point_mean,
             batch_mode, div, scale, _p(grad))
        return grad, None, None, None, None, None, (grad_out if ctx.needs_input_grad[6] else None)   # d(out)/d(add) = 1


def chamfer_reduce(cham, lengths, point_reduction, batch_reduction, scale=1.0, add=None):
    """pytorch3d_chamfer.py:295-326 in one launch: cham [N,P] (rows beyond a cloud's length already zero) -> sum or mean over
    the points (mean divides by lengths [N] i64), then None / sum / mean over the batch, times the constant `scale`.
    add: a device scalar (a running total of loss terms) added to a batch-reduced result inside the same launch."""
    _need_hip(cham, lengths, add)
    N = cham.shape[0]
    batch_mode = {None: 0, "sum": 1, "mean": 2}[batch_reduction]
    if add is not None and (batch_mode == 0 or add.numel() != 1 or add.dtype != torch.float32):
        raise ValueError("add must be a float32 scalar and needs a batch reduction")
    return _ChamferReduce.apply(_f32(cham), _i64(lengths), point_reduction == "mean", batch_mode, float(N), float(scale), add)


class _PoseOutput(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, raw, weight_orient):
        n = pos.numel() // 3
        out = torch.empty((n, 6), dtype=torch.float32, device=pos.device)
        _run("pose_output", pos, _lib.load().mp_pose_output_f32, _p(pos), _p(raw), n, float(weight_orient), _p(out))
        ctx.save_for_backward(raw)
        ctx.meta = (n, float(weight_orient), tuple(pos.shape), tuple(raw.shape))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (raw,) = ctx.saved_tensors
        n, w, pshape, rshape = ctx.meta
        grad_out = _f32(grad_out)
        gp = torch.empty(pshape, dtype=torch.float32, device=raw.device) if ctx.needs_input_grad[0] else None
        gr = torch.empty(rshape, dtype=torch.float32, device=raw.device) if ctx.needs_input_grad[1] else None
        _run("pose_output_bwd", raw, _lib.load().mp_pose_output_bwd_f32, _p(grad_out), _p(raw), n, w, _p(gp), _p(gr))
        return gp, gr, None


def pose_output(pos, raw, weight_orient):
    """models/pointnet2_cls_ssg.py:332-339 in one launch: pos, raw [B, n_pose*3] -> [B*n_pose, 6] rows
    (position, weight_orient * unit(tanh(raw)))."""
    _need_hip(pos, raw)
    if pos.shape != raw.shape or pos.numel() % 3:
        raise ValueError("pose_output: positions and raw normals must have the same shape, a multiple of 3 values")
    return _PoseOutput.apply(_f32(pos), _f32(raw), float(weight_orient))


class _MaskLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, scores, target_ids, match, uniq, w_masks, w_conf, no_stroke_weight, status, add):
        B, M, S = pred.shape
        dev = pred.device
        per_mask = torch.empty((B * M,), dtype=torch.float32, device=dev)
        out = torch.empty((), dtype=torch.float32, device=dev)
        n_matched = torch.empty((1,), dtype=torch.float32, device=dev)
        _run("mask_loss", pred, _lib.load().mp_mask_loss_f32, _p(pred), _p(scores), _p(target_ids), _p(match), _p(uniq), B, M, S,
             float(w_masks), float(w_conf), float(no_stroke_weight), _p(per_mask), _p(out), _p(n_matched), _p(status), _p(add))
        ctx.save_for_backward(pred, scores, target_ids, match, uniq, n_matched)
        ctx.meta = (B, M, S, float(w_masks), float(w_conf), float(no_stroke_weight))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        pred, scores, target_ids, match, uniq, n_matched = ctx.saved_tensors
        B, M, S, w_masks, w_conf, nsw = ctx.meta
        grad_out = _f32(grad_out)
        gm = torch.empty_like(pred)
        gs = torch.empty_like(scores) if ctx.needs_input_grad[1] else None
        _run("mask_loss_bwd", pred, _lib.load().mp_mask_loss_bwd_f32, _p(grad_out), _p(pred), _p(scores), _p(target_ids), _p(match),
             _p(uniq), _p(n_matched), B, M, S, w_masks, w_conf, nsw, _p(gm), _p(gs))
        return gm, gs, None, None, None, None, None, None, None, (grad_out if ctx.needs_input_grad[9] else None)


def mask_loss(pred_masks, scores, target_ids, match, uniq, w_masks, w_conf, no_stroke_weight, status=None, add=None):
    """loss_handler.py:877-934 (binary targets) after mask_match: w_masks * matched-BCE.sum(-1).mean() +
    w_conf * weighted confidence BCE.mean(), forward in two launches, backward in one.  status: mask_match's per-sample
    status i32 [B]; a non-zero entry (conditions the reference asserts on / scipy raises for) makes the loss NaN."""
    _need_hip(pred_masks, scores, target_ids, match, uniq, status)
    if status is not None and (status.dtype != torch.int32 or not status.is_contiguous()):
        status = status.to(torch.int32).contiguous()
    if add is not None and (add.numel() != 1 or add.dtype != torch.float32):
        raise ValueError("add must be a float32 scalar")
    return _MaskLoss.apply(_f32(pred_masks), _f32(scores), _f32(target_ids), _i64(match), _f32(uniq), w_masks, w_conf,
                           no_stroke_weight, status, add)


class _BnReluRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, drop_p, rng, layer):
        B, C = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((2, C), dtype=torch.float32, device=x.device)
        if rng is not None:    # the nn.Dropout behind the block in the same launch
            _run("bn_relu_rows", x, _lib.load().mp_bn_relu_drop_rows_f32, _p(x), B, C, int(training), float(momentum), float(eps), _p(gamma),
                 _p(beta), _p(running_mean), _p(running_var), _p(y), stats[0].data_ptr(), stats[1].data_ptr(), float(drop_p), _p(rng), int(layer))
        else:
            _run("bn_relu_rows", x, _lib.load().mp_bn_relu_rows_f32, _p(x), B, C, int(training), float(momentum), float(eps), _p(gamma),
                 _p(beta), _p(running_mean), _p(running_var), _p(y), stats[0].data_ptr(), stats[1].data_ptr())
        ctx.save_for_backward(x, y, gamma, stats)
        ctx.training = bool(training)
        ctx.drop_p = float(drop_p) if rng is not None else None
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, y, gamma, stats = ctx.saved_tensors
        B, C = x.shape
        grad_y = _f32(grad_y)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gg = torch.empty((C,), dtype=torch.float32, device=x.device) if (gamma is not None and ctx.needs_input_grad[1]) else None
        gb = torch.empty((C,), dtype=torch.float32, device=x.device) if ctx.needs_input_grad[2] else None
        if ctx.drop_p is not None:
            _run("bn_relu_rows_bwd", x, _lib.load().mp_bn_relu_drop_rows_bwd_f32, _p(grad_y), _p(y), _p(x), B, C, int(ctx.training), _p(gamma),
                 stats[0].data_ptr(), stats[1].data_ptr(), _p(gx), _p(gg), _p(gb), ctx.drop_p)
        else:
            _run("bn_relu_rows_bwd", x, _lib.load().mp_bn_relu_rows_bwd_f32, _p(grad_y), _p(y), _p(x), B, C, int(ctx.training), _p(gamma),
                 stats[0].data_ptr(), stats[1].data_ptr(), _p(gx), _p(gg), _p(gb))
        return gx, gg, gb, None, None, None, None, None, None, None, None


def bn_relu_rows(x, bn, dropout=None):
    """F.relu(bn(x)) for an nn.BatchNorm1d `bn` and x [B, C] with a small batch (models/pointnet2_cls_ssg.py:309-327): one
    launch forward, one backward.  Updates running statistics like the module does; the caller advances
    `num_batches_tracked` (heads batch that into one launch).
    dropout = (p, rng, layer): the nn.Dropout(p) that follows the block, in the same launch -- rng is a device int64 [2] tensor
    (seed, step) whose step the caller advances once per training step; `layer` separates the blocks of one step."""
    _need_hip(x)
    if x.ndim != 2 or x.shape[1] != bn.num_features:
        raise ValueError("bn_relu_rows expects [B, C] input matching the BatchNorm width")
    training = bn.training or bn.running_mean is None
    track = bn.track_running_stats and bn.running_mean is not None
    momentum = bn.momentum if bn.momentum is not None else 1.0 / max(float(bn.num_batches_tracked), 1.0)
    p, rng, layer = dropout if dropout is not None else (0.0, None, 0)
    if rng is not None:
        _need_hip(rng)
        if rng.dtype != torch.int64 or rng.numel() != 2:
            raise ValueError("dropout rng must be an int64 tensor (seed, step)")
    y = _BnReluRows.apply(_f32(x), bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None,
                          training, momentum, bn.eps, p, rng, layer)
    if RELU_TAP is not None:
        RELU_TAP.append(y.detach() > 0)
    return y

