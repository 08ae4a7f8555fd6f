"""Shared MLP + BatchNorm + ReLU + max-pool of a set-abstraction level (models/pointnet2_utils.py:208-214) on the
fused gfx950 path (csrc/sa_mlp.hip): one C-ABI call forward, one backward.

Input is the grouped tensor in positions-major layout [B,S,K,C] (what ops.group produces); a 1x1 Conv2d over
[B,C,K,S] is a GEMM over the last axis and BatchNorm2d statistics are statistics over all B*S*K rows, so the module
parameters (Conv2d weight [Co,Ci,1,1], BatchNorm2d weight/bias/running stats) are consumed as they are and the
reference's state_dict stays valid.
"""
import ctypes
import weakref

import os

import torch

from . import _lib, ops


def _ptr(t):
    return None if t is None else t.data_ptr()


def _r16(t):
    """The values a bf16 matrix-core operand carries (nearest even), in fp32 storage."""
    return t.detach().to(torch.bfloat16).to(torch.float32)


ROUNDED_INPUTS = {}     # data_ptr of grouped coordinate rows whose values are bf16 values already (rounded with the sampling plan: harness)


def rounds_first_input(convs, K, dtype):
    """True when _SharedMLPMax.forward rounds its input rows to bf16 itself (the bf16 variant's recomputed first layer on a chain
    with bf16 activation storage): what a producer may do ahead of time instead (ROUNDED_INPUTS)."""
    if dtype != "bf16":
        return False
    cin = convs[0].in_channels
    widths = [c.out_channels for c in convs]
    if WIDEN_INTERIOR:        # (as _widen_interior: interior widths between 64 and 128 run as 128)
        widths = [128 if (i < len(widths) - 1 and 64 < c < 128) else c for i, c in enumerate(widths)]
    chans = [(cin + 3) // 4 * 4] + widths
    lib = _lib.load()
    ch = (ctypes.c_int64 * len(chans))(*chans)
    n = len(convs)
    return bool(lib.mp_sa_mlp_recompute_first(n, ch, K)) and bool(lib.mp_sa_mlp_bf16_storage(n, ch, K, 1))


_R16_LAST = [None, None]     # (tensor, its rounded copy) of the last feature table rounded by _r16_shared


def _r16_shared(t):
    """_r16 of a feature table that several scales of one multi-scale level round in turn: computed once (two launches) and handed out
    again while the SAME tensor object comes back unmodified (the reference held here keeps its storage from being reused)."""
    last = _R16_LAST[0]
    if last is not None and last[0] is t and last[1] == t._version:
        return _R16_LAST[1]
    r = _r16(t)
    _R16_LAST[0], _R16_LAST[1] = (t, t._version), r
    return r


BN_SLOTS = 8      # MP_BN_SLOTS
BN_FUSED = os.environ.get("MASKPLANNER_BN_FUSED", "1") != "0"   # 0: BatchNorm finalize as launches of its own (A/B, debugging)


def bn_state(bn, C):
    """The persistent, zero-initialised scratch of a BatchNorm layer for the library's consumer-side finalize (mp_mlp_layer_t::bn_state:
    MP_BN_STATE_DOUBLES(C) doubles): kept with the module, one per (width, device); the library leaves it zero after every call.
    One call at a time per module (two streams running the SAME module concurrently would share it)."""
    dev = bn.weight.device
    cache = bn.__dict__.setdefault("_mp_bn_state", {})
    t = cache.get((C, dev))
    if t is None:
        t = cache[(C, dev)] = torch.zeros(4 * BN_SLOTS * C, dtype=torch.float64, device=dev)
    return t


def reset_bn_state(module):
    """Zero every cached BatchNorm scratch under `module` (after an aborted step: a kernel that failed between a producer and its
    consumer leaves partial sums behind)."""
    for m in module.modules():
        for t in m.__dict__.get("_mp_bn_state", {}).values():
            t.zero_()


# Test hook (tests/test_gpu_routing.py): a list that receives, per level, (the max-pool arg-max [G, C] -- member index inside the group --,
# [(raw Z_l [P, C] | None, scale [C], shift [C]) per layer]) as the forward pass computes them, so that an oracle can be evaluated with
# the SAME discrete decisions (pool routing, ReLU masks = (Z * scale + shift > 0)).
ROUTE_TAP = None


# None: every level advances its BatchNorm counters (num_batches_tracked) itself.  A list: the counters are collected here
# instead and the owner (harness.TrainStep) advances all of them -- set-abstraction levels and heads -- in one launch per step.
DEFERRED_TICKS = None


def flush_ticks():
    if DEFERRED_TICKS:
        torch._foreach_add_(list(DEFERRED_TICKS), 1)
        del DEFERRED_TICKS[:]


class _SharedMLPMax(torch.autograd.Function):
    """args: x [P, C0] (contiguous), K, training, momentum, eps, L, then per layer:
    weight[Co,Ci], bias|None, gamma, beta, running_mean|None, running_var|None."""

    @staticmethod
    def forward(ctx, x, grad_to, K, training, momentum, eps, n_layers, grad_cols, bf16, sync_group, states, *params):
        dev = x.device
        P, C0 = x.shape
        layers = (_lib.MlpLayer * n_layers)()
        keep = []  # tensors the structs point to: alive until the call returns; saved ones also for backward
        chans = [C0] + [params[6 * l].shape[0] for l in range(n_layers)]
        lib = _lib.load()
        ch = (ctypes.c_int64 * len(chans))(*chans)
        # first layer of a level fed by bare coordinates (4 input channels): Z_0 is recomputed by its consumers instead of being
        # written once and read three times (sa_mlp.hip, SRC_*_RC) -- no buffer for it at all
        recompute_first = (not x.requires_grad) and bool(lib.mp_sa_mlp_recompute_first(n_layers, ch, K))
        # bf16 variant: chains that run entirely on the position-stream kernels keep Z_l in memory as bf16 (mp_sa_mlp_bf16_storage);
        # the others keep fp32 storage and store their first layer
        store16 = bool(bf16) and recompute_first and bool(lib.mp_sa_mlp_bf16_storage(n_layers, ch, K, 1))
        if bf16 and not store16:
            recompute_first = False
        if bf16 and recompute_first:
            # the recomputed first layer of the bf16 variant is an exact product of ROUNDED operands: the kernels get x and W_0 as
            # bf16 values (in fp32 storage); the gradient still goes to the unrounded parameter
            if x.data_ptr() not in ROUNDED_INPUTS:
                x = _r16(x)
            params = (_r16(params[0]),) + tuple(params[1:])
        for l in range(n_layers):
            w, b, gam, bet, rm, rv = params[6 * l:6 * l + 6]
            co, ci = w.shape
            skip_z = l == 0 and recompute_first
            z = None if skip_z else torch.empty((P, co), dtype=torch.bfloat16 if store16 else torch.float32, device=dev)
            stats = torch.empty((4, co), dtype=torch.float32, device=dev)  # mean, rstd, scale, shift
            st = states[l] if states is not None else None
            keep.append((w, b, gam, bet, rm, rv, z, stats, st))
            layers[l] = _lib.MlpLayer(_ptr(w), _ptr(b), _ptr(gam), _ptr(bet), _ptr(rm), _ptr(rv), ci, co, _ptr(z),
                                      stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), _ptr(st))
        G = P // K
        cl = chans[-1]
        out = torch.empty((G, cl), dtype=torch.float32, device=dev)
        argk = torch.empty((G, cl), dtype=torch.int32, device=dev)
        zmax = torch.empty((G, cl), dtype=torch.float32, device=dev)
        ws = torch.empty((lib.mp_sa_mlp_workspace_bytes(P, K, n_layers, ch, 0),), dtype=torch.uint8, device=dev)
        if sync_group is not False and training:
            # global-batch BatchNorm statistics: the library calls back once per layer to all-reduce its fp64 sums (sync_bn.py)
            from . import sync_bn
            ex = sync_bn.Exchange(sync_group, max(chans), dev)
            ops._run("sa_mlp_fwd", x, lib.mp_sa_mlp_fwd_ex, _ptr(x), P, K, n_layers, layers, int(training), float(momentum),
                     float(eps), _ptr(out), _ptr(argk), _ptr(zmax), _ptr(ws), ws.numel(), int(bool(bf16)), ctypes.byref(ex.struct))
            if ex.error is not None:
                raise ex.error
        else:
            sync_group = False
            ops._run("sa_mlp_fwd", x, lib.mp_sa_mlp_fwd_bf16 if bf16 else lib.mp_sa_mlp_fwd_f32, _ptr(x), P, K, n_layers, layers, int(training), float(momentum),
                     float(eps), _ptr(out), _ptr(argk), _ptr(zmax), _ptr(ws), ws.numel())
        ctx.sync_group = sync_group
        ctx.meta = (P, K, bool(training), n_layers, chans, int(grad_cols), bool(bf16))
        ctx.compact = grad_to is not None and grad_cols > 0 and grad_cols % 4 == 0 and grad_to.numel() == P * grad_cols
        ctx.grad_to_shape = None if grad_to is None else tuple(grad_to.shape)
        if grad_to is not None and not ctx.compact:
            raise ValueError("grad_to needs a feats-first input whose feature width is a multiple of 4")
        ctx.keep = keep
        if ROUTE_TAP is not None:
            ROUTE_TAP.append((argk, [(k[6], k[7][2], k[7][3]) for k in keep]))
        ctx.save_for_backward(x, out, argk, zmax)
        ctx.mark_non_differentiable(argk, zmax)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, out, argk, zmax = ctx.saved_tensors
        P, K, training, n_layers, chans, grad_cols, bf16 = ctx.meta
        dev = x.device
        grad_out = grad_out.contiguous().float()
        layers = (_lib.MlpLayer * n_layers)()
        grads = (_lib.MlpGrads * n_layers)()
        ret = []
        # the weight gradients of a level share one allocation, in layer order: the library then clears them with a single
        # launch (they are accumulated with atomics) instead of one per layer
        dw_all = torch.empty((sum(k[0].numel() for k in ctx.keep),), dtype=torch.float32, device=dev)
        dw_off = 0
        for l, (w, b, gam, bet, rm, rv, z, stats, st) in enumerate(ctx.keep):
            co, ci = w.shape
            layers[l] = _lib.MlpLayer(_ptr(w), _ptr(b), _ptr(gam), _ptr(bet), _ptr(rm), _ptr(rv), ci, co, _ptr(z),
                                      stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), _ptr(st))
            dw = dw_all[dw_off:dw_off + w.numel()].view_as(w)
            dw_off += w.numel()
            db = None if b is None else torch.empty_like(b)
            dg, dbe = torch.empty_like(gam), torch.empty_like(bet)
            grads[l] = _lib.MlpGrads(_ptr(dw), _ptr(db), _ptr(dg), _ptr(dbe))
            ret += [dw, db, dg, dbe, None, None]
        # with grad_cols only the leading (feature) columns carry a gradient; the library writes zeros into the others -- or, when the
        # caller differentiates the FEATURES themselves (grad_to: x was assembled from them without autograd), nothing but those columns:
        # a compact [P, grad_cols] gradient (grad_x0_cols < 0), no clear of the rest, no slice / copy downstream
        compact = ctx.compact
        if compact:
            gx = torch.empty((P, grad_cols), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
            grad_cols = -grad_cols
        else:
            gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ch = (ctypes.c_int64 * len(chans))(*chans)
        lib = _lib.load()
        ws = torch.empty((lib.mp_sa_mlp_workspace_bytes(P, K, n_layers, ch, 1),), dtype=torch.uint8, device=dev)
        if ctx.sync_group is not False:
            from . import sync_bn
            ex = sync_bn.Exchange(ctx.sync_group, max(chans), dev)
            ops._run("sa_mlp_bwd", x, lib.mp_sa_mlp_bwd_ex, _ptr(x), P, K, n_layers, layers, int(training), _ptr(grad_out),
                     _ptr(out), _ptr(argk), _ptr(zmax), grads, _ptr(gx), grad_cols, _ptr(ws), ws.numel(), int(bool(bf16)),
                     ctypes.byref(ex.struct))
            if ex.error is not None:
                raise ex.error
        else:
            ops._run("sa_mlp_bwd", x, lib.mp_sa_mlp_bwd_bf16 if bf16 else lib.mp_sa_mlp_bwd_f32, _ptr(x), P, K, n_layers, layers, int(training), _ptr(grad_out),
                     _ptr(out), _ptr(argk), _ptr(zmax), grads, _ptr(gx), grad_cols, _ptr(ws), ws.numel())
        ctx.keep = None
        if compact:
            return (None, None if gx is None else gx.view(ctx.grad_to_shape), None, None, None, None, None, None, None, None, None, *ret)
        return (gx, None, None, None, None, None, None, None, None, None, None, *ret)


class _SharedMLPMaxFactored(torch.autograd.Function):
    """The level with its first layer factorised (sa_mlp.hip, first_factored_fwd_kernel): Z_0[p] = A[b, idx[p]] + W_x (x[idx[p]] - c),
    A = F W_f^T computed per SOURCE point by the caller.  args: A [B,N,Co] (carries the gradient), xyz [B,N,3], new_xyz [B,S,3],
    idx [B,S,K] i64, training, momentum, eps, L, then the layers' parameters as for _SharedMLPMax with layer 0 = (W_x | 0) [Co, 4].
    Backward: the library writes dZ_0 [P, Co + 4]; its reduction over the gathering rows (ops.group's backward kernel) is dA."""

    @staticmethod
    def forward(ctx, A, xyz, new_xyz, idx, training, momentum, eps, n_layers, bf16, sync_group, states, *params):
        dev = A.device
        B, N, C0 = A.shape
        _, S, K = idx.shape
        P = B * S * K
        layers = (_lib.MlpLayer * n_layers)()
        keep = []
        chans = [4] + [params[6 * l].shape[0] for l in range(n_layers)]
        lib = _lib.load()
        ch_ = (ctypes.c_int64 * len(chans))(*chans)
        sync = sync_group is not False and training      # global-batch BatchNorm statistics (sync_bn.py), as in _SharedMLPMax
        store16 = bool(bf16) and bool(lib.mp_sa_mlp_bf16_storage(n_layers, ch_, K, 2))       # (see _SharedMLPMax)
        for l in range(n_layers):
            w, b, gam, bet, rm, rv = params[6 * l:6 * l + 6]
            co, ci = w.shape
            z = torch.empty((P, co), dtype=torch.bfloat16 if store16 else torch.float32, device=dev)
            stats = torch.empty((4, co), dtype=torch.float32, device=dev)
            st = states[l] if states is not None else None
            keep.append((w, b, gam, bet, rm, rv, z, stats, st))
            layers[l] = _lib.MlpLayer(_ptr(w), _ptr(b), _ptr(gam), _ptr(bet), _ptr(rm), _ptr(rv), ci, co, _ptr(z),
                                      stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), _ptr(st))
        G = P // K
        cl = chans[-1]
        out = torch.empty((G, cl), dtype=torch.float32, device=dev)
        argk = torch.empty((G, cl), dtype=torch.int32, device=dev)
        zmax = torch.empty((G, cl), dtype=torch.float32, device=dev)
        ch = (ctypes.c_int64 * len(chans))(*chans)
        ws = torch.empty((lib.mp_sa_mlp_workspace_bytes(P, K, n_layers, ch, 0),), dtype=torch.uint8, device=dev)
        g = _lib.Gather(_ptr(A), _ptr(xyz), _ptr(new_xyz), _ptr(idx), N, S, C0, None)
        if sync:
            from . import sync_bn
            ex = sync_bn.Exchange(sync_group, max(chans), dev)
            ops._run("sa_mlp_fwd", A, lib.mp_sa_mlp_fwd_gather_ex, ctypes.byref(g), P, K, n_layers, layers, int(training), float(momentum),
                     float(eps), _ptr(out), _ptr(argk), _ptr(zmax), _ptr(ws), ws.numel(), int(bool(bf16)), ctypes.byref(ex.struct))
            if ex.error is not None:
                raise ex.error
        else:
            sync_group = False
            ops._run("sa_mlp_fwd", A, lib.mp_sa_mlp_fwd_gather_bf16 if bf16 else lib.mp_sa_mlp_fwd_gather_f32, ctypes.byref(g), P, K, n_layers, layers, int(training), float(momentum),
                     float(eps), _ptr(out), _ptr(argk), _ptr(zmax), _ptr(ws), ws.numel())
        ctx.sync_group = sync_group
        ctx.meta = (P, K, bool(training), n_layers, chans, (B, N, S, C0), bool(bf16))
        ctx.keep = keep
        if ROUTE_TAP is not None:
            ROUTE_TAP.append((argk, [(k[6], k[7][2], k[7][3]) for k in keep]))
        ctx.save_for_backward(A, xyz, new_xyz, idx, out, argk, zmax)
        ctx.mark_non_differentiable(argk, zmax)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        A, xyz, new_xyz, idx, out, argk, zmax = ctx.saved_tensors
        P, K, training, n_layers, chans, (B, N, S, C0), bf16 = ctx.meta
        dev = A.device
        ex = None
        if ctx.sync_group is not False:
            from . import sync_bn
            ex = sync_bn.Exchange(ctx.sync_group, max(chans), dev)

        def run_bwd(lib, *a):
            if ex is not None:
                ops._run("sa_mlp_bwd", A, lib.mp_sa_mlp_bwd_gather_ex, *a, int(bool(bf16)), ctypes.byref(ex.struct))
                if ex.error is not None:
                    raise ex.error
            else:
                ops._run("sa_mlp_bwd", A, lib.mp_sa_mlp_bwd_gather_bf16 if bf16 else lib.mp_sa_mlp_bwd_gather_f32, *a)
        grad_out = grad_out.contiguous().float()
        layers = (_lib.MlpLayer * n_layers)()
        grads = (_lib.MlpGrads * n_layers)()
        ret = []
        dw_all = torch.empty((sum(k[0].numel() for k in ctx.keep),), dtype=torch.float32, device=dev)
        dw_off = 0
        for l, (w, b, gam, bet, rm, rv, z, stats, st) in enumerate(ctx.keep):
            co, ci = w.shape
            layers[l] = _lib.MlpLayer(_ptr(w), _ptr(b), _ptr(gam), _ptr(bet), _ptr(rm), _ptr(rv), ci, co, _ptr(z),
                                      stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), _ptr(st))
            dw = dw_all[dw_off:dw_off + w.numel()].view_as(w)
            dw_off += w.numel()
            db = None if b is None else torch.empty_like(b)
            dg, dbe = torch.empty_like(gam), torch.empty_like(bet)
            grads[l] = _lib.MlpGrads(_ptr(dw), _ptr(db), _ptr(dg), _ptr(dbe))
            ret += [dw, db, dg, dbe, None, None]
        stride = C0 + 4
        ch = (ctypes.c_int64 * len(chans))(*chans)
        lib = _lib.load()
        ws = torch.empty((lib.mp_sa_mlp_workspace_bytes(P, K, n_layers, ch, 1),), dtype=torch.uint8, device=dev)
        rows = CSR_ROWS.get(idx.data_ptr())           # sorted row lists prepared with the sampling plan (harness), else the library sorts
        if rows is not None and (rows.numel() != 2 * B * S * K or rows.dtype != torch.int32):
            rows = None
        g = _lib.Gather(_ptr(A), _ptr(xyz), _ptr(new_xyz), _ptr(idx), N, S, C0, _ptr(rows))
        # dA reduced inside the library from rows sorted by source point (no dZ_0 round trip; summation order not fixed), unless a
        # deterministic run asks for the ordered scatter: then dZ_0 comes back and ops.group's backward kernel reduces it
        fused = FACTORED_REDUCE and not ops.DETERMINISTIC and N <= 15000 and S * K < (1 << 24)
        if fused:
            gA = ops.zeroed_empty((B, N, C0), torch.float32, dev)
            run_bwd(lib, ctypes.byref(g), P, K, n_layers, layers, int(training), _ptr(grad_out),
                    _ptr(out), _ptr(argk), _ptr(zmax), grads, _ptr(gA), 0, _ptr(ws), ws.numel())
            ctx.keep = None
            return (gA, None, None, None, None, None, None, None, None, None, None, *ret)
        gz = torch.empty((P, stride), dtype=torch.float32, device=dev)       # dZ_0 rows (the pad quad is never read)
        run_bwd(lib, ctypes.byref(g), P, K, n_layers, layers, int(training), _ptr(grad_out),
                _ptr(out), _ptr(argk), _ptr(zmax), grads, _ptr(gz), C0, _ptr(ws), ws.numel())
        ctx.keep = None
        gA = None
        if ctx.needs_input_grad[0]:
            gA = torch.empty((B, N, C0), dtype=torch.float32, device=dev)
            ops._run("group_bwd", gz, lib.mp_group_bwd_f32, _ptr(gz), _ptr(idx), B, N, S, K, C0, 1, stride, _ptr(gA), int(ops.DETERMINISTIC))
        return (gA, None, None, None, None, None, None, None, None, None, None, *ret)


# idx.data_ptr() -> int32 [2, B, S*K]: the sorted row lists (mp_csr_rows_i64) of a level's ball-query result, registered by a harness that
# computes them with the sampling plan (they depend on idx alone); the factorised backward then launches no sort of its own
CSR_ROWS = {}


def csr_rows(idx, N, out=None):
    """idx i64 [B, S, K] (source points < N) -> int32 [2, B, S*K]: rows sorted by source point, and that point (mp_csr_rows_i64)."""
    B, S, K = idx.shape
    if out is None:
        out = torch.empty((2, B, S * K), dtype=torch.int32, device=idx.device)
    ops._run("csr_rows", idx, _lib.load().mp_csr_rows_i64, _ptr(idx), B, N, S * K, _ptr(out))
    return out


PER_POINT_DW_SLICES = 8      # K slices per cloud of the batched weight-gradient GEMM below


_ZERO_COLS = {}


def _zero_col(n, device):
    t = _ZERO_COLS.get((n, device))
    if t is None:
        t = _ZERO_COLS[(n, device)] = torch.zeros((n, 1), dtype=torch.float32, device=device)
    return t


class _PerPointFirst(torch.autograd.Function):
    """The feature half of a factorised first layer and the split of its weight, with a backward that costs four launches:
    (feats [B,N,CF], w [Co,Cin] in the module's column order, xyz_first) -> A = feats W_f^T [B,N,Co], (W_x | 0) [Co,4].
    Backward: dF = dA W_f (one GEMM), dW_f = sum_b dA_b^T F_b as a BATCHED GEMM over the clouds + a sum (the flat [Co, B*N] x
    [B*N, CF] form sends rocBLAS to a 16-tile kernel with a 16 384-long K loop: 93 us at the bench shape), dW assembled with one cat."""

    @staticmethod
    def forward(ctx, feats, w, xyz_first, bf16=False):
        CF = feats.shape[2]
        if bf16:       # the bf16 variant: both operands of the per-point map rounded (exact products, fp32 sums); W_x is rounded by the kernels
            feats, w = _r16_shared(feats), torch.cat([w[:, :3], _r16(w[:, 3:])], 1) if xyz_first else torch.cat([_r16(w[:, :CF]), w[:, CF:]], 1)
        wx, wf = (w[:, :3], w[:, 3:]) if xyz_first else (w[:, CF:], w[:, :CF])
        A = torch.matmul(feats, wf.t())
        wx4 = torch.cat([wx, _zero_col(wx.shape[0], wx.device)], 1)       # (one launch: F.pad is a fill + a copy)
        ctx.save_for_backward(feats, w)
        ctx.xyz_first = bool(xyz_first)
        return A, wx4

    @staticmethod
    def backward(ctx, gA, gwx4):
        feats, w = ctx.saved_tensors
        CF = feats.shape[2]
        wf = w[:, 3:] if ctx.xyz_first else w[:, :CF]
        gfeats = torch.matmul(gA, wf) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            # (more, shorter K slices than clouds: one 128 x 128 tile per cloud leaves the GEMM on 32 workgroups with a 512-long K loop)
            B, N, _ = feats.shape
            Co = gA.shape[2]
            if (gA.is_cuda and gA.dtype == torch.float32 and gA.is_contiguous() and feats.is_contiguous() and not ops.DETERMINISTIC
                    and Co % 4 == 0 and CF % 4 == 0 and Co <= 1024 and CF <= 1024):
                # [r4] the library's row-sliced weight-gradient GEMM (split planes, atomics between the slices): one launch instead of the
                # batched GEMM + the 16 MB sum over its slices (29 us at the bench shape)
                gwf = ops.zeroed_empty((Co, CF), torch.float32, gA.device)
                ops._run("dw_gemm", gA, _lib.load().mp_dw_gemm_f32, _ptr(gA), _ptr(feats), B * N, Co, CF, _ptr(gwf))
            else:
                sl = next(d for d in (PER_POINT_DW_SLICES, 4, 2, 1) if N % d == 0)
                gwf = torch.bmm(gA.reshape(B * sl, N // sl, -1).transpose(1, 2), feats.reshape(B * sl, N // sl, CF)).sum(0)
            gwx = gwx4[:, :3] if gwx4 is not None else torch.zeros_like(w[:, :3])
            gw = torch.cat([gwx, gwf], dim=1) if ctx.xyz_first else torch.cat([gwf, gwx], dim=1)
        return gfeats, gw, None, None


# MASKPLANNER_FACTORED_FIRST: which levels with input features run their first layer factorised (linear map per source point, then
# a gather-add) instead of as a GEMM over the grouped rows.  "1" (default): every such level -- the bench's second level (2.56 vs
# 2.64 ms per step) and the multi-scale levels, whose 323-input first layers are otherwise tiled GEMMs over 935 MB of grouped rows
# (config 5: 11.5 -> 9.0 ms); "msg": the multi-scale levels only; "0": none (the grouped route).
FACTORED_FIRST = os.environ.get("MASKPLANNER_FACTORED_FIRST", "1")
FACTORED_REDUCE = True   # False (tests): dZ_0 written out and reduced by the grouping backward's kernel, as ops.DETERMINISTIC does


def factored_supported(feats, K, convs, bns, dtype="f32", sync_bn=None):
    """True when shared_mlp_max_factored can take this level (otherwise group + shared_mlp_max)."""
    from .sync_bn import resolve
    if dtype not in ("f32", "bf16") or feats is None or not feats.is_cuda or feats.dtype != torch.float32:
        return False        # ([r3] SyncBN levels qualify too: mp_sa_mlp_{fwd,bwd}_gather_ex)
    if len(convs) < 2 or convs[0].in_channels != feats.shape[2] + 3 or convs[0].out_channels not in (64, 128, 256):
        return False
    return all(c.out_channels % 4 == 0 for c in convs)


def shared_mlp_max_factored(xyz, feats, new_xyz, idx, convs, bns, weight_order="xyz_first", dtype="f32", sync_bn=None):
    """The level's output [B,S,Cout] from xyz [B,N,3], feats [B,N,CF], new_xyz [B,S,3], idx [B,S,K] without a grouped tensor and
    without a first-layer GEMM over the grouped rows.  weight_order: where the coordinate columns sit in the first conv's weight --
    "xyz_first" (PointNetSetAbstraction, models/pointnet2_utils.py:138) or "xyz_last" (the multi-scale class, :262).
    Check factored_supported() first."""
    import torch.nn.functional as F
    ops._need_hip(xyz, feats, new_xyz, idx)
    B, S, K = idx.shape
    CF = feats.shape[2]
    training = bns[0].training
    params = []
    for i, (conv, bn) in enumerate(zip(convs, bns)):
        if bn.training != training:
            raise ValueError("all BatchNorm layers of a set-abstraction level must share one mode")
        w = conv.weight.view(conv.out_channels, conv.in_channels)
        if i == 0:
            # A [B, N, Co]: the feature part, once per source point; w: (W_x | 0) [Co, 4]
            A, w = _PerPointFirst.apply(feats.contiguous(), w, weight_order == "xyz_first", dtype == "bf16")
        track = bn.track_running_stats and bn.running_mean is not None
        if not training and not track:
            raise NotImplementedError("eval-mode BatchNorm without running statistics")
        params += [w, conv.bias, bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None]
    if training:
        counters = [bn.num_batches_tracked for bn in bns if bn.track_running_stats and bn.num_batches_tracked is not None]
        if counters:
            if DEFERRED_TICKS is not None:
                DEFERRED_TICKS.extend(counters)
            else:
                torch._foreach_add_(counters, 1)
    bn0 = bns[0]
    momentum = bn0.momentum if bn0.momentum is not None else 1.0 / max(float(bn0.num_batches_tracked), 1.0)
    from .sync_bn import resolve
    sync_group = resolve(sync_bn)
    writeback = _widen_interior(params, [c.out_channels for c in convs]) if (sync_group is False and WIDEN_INTERIOR) else []
    states = [bn_state(bn, params[6 * i].shape[0]) for i, bn in enumerate(bns)] if (training and sync_group is False and BN_FUSED) else None
    out = _SharedMLPMaxFactored.apply(A.contiguous(), xyz.contiguous().float(), new_xyz.contiguous().float(), idx.contiguous(), training,
                                      momentum, bn0.eps, len(convs), dtype == "bf16", sync_group, states, *params)
    if training:
        for dst, src in writeback:
            dst.copy_(src[:dst.numel()])
    return out.view(B, S, -1)


class _PermuteCols(torch.autograd.Function):
    """dst[:, c] = src[:, perm[c]] (zero where perm[c] < 0); perm / inv are cached int32 device tensors."""

    @staticmethod
    def forward(ctx, src, perm, inv):
        R, Cs = src.shape
        dst = torch.empty((R, perm.numel()), dtype=torch.float32, device=src.device)
        ops._run("permute_cols", src, _lib.load().mp_permute_cols_f32, _ptr(src), _ptr(perm), R, Cs, perm.numel(), _ptr(dst))
        ctx.save_for_backward(perm, inv)
        return dst

    @staticmethod
    def backward(ctx, grad):
        perm, inv = ctx.saved_tensors
        grad = grad.contiguous()
        R = grad.shape[0]
        out = torch.empty((R, inv.numel()), dtype=torch.float32, device=grad.device)
        ops._run("permute_cols", grad, _lib.load().mp_permute_cols_f32, _ptr(grad), _ptr(inv), R, perm.numel(), inv.numel(), _ptr(out))
        return out, None, None


class _PermuteColsMulti(torch.autograd.Function):
    """_PermuteCols for several weights at once: one launch forward, one launch for all the gradients in backward.
    args: n, then per weight (src, perm, inv)."""

    @staticmethod
    def forward(ctx, n, *args):
        srcs, perms, invs = args[0::3], args[1::3], args[2::3]
        dsts = [torch.empty((s_.shape[0], p_.numel()), dtype=torch.float32, device=s_.device) for s_, p_ in zip(srcs, perms)]
        _permute_multi(srcs, perms, [s_.shape[1] for s_ in srcs], dsts)
        ctx.save_for_backward(*perms, *invs)
        ctx.n = n
        # an output no level picked up (the factorised first layer reads its weight in the module's own column order) must come back as
        # None, not as a materialised zero: that was a fill, a wasted row of the gradient launch and an add onto the real gradient
        ctx.set_materialize_grads(False)
        return tuple(dsts)

    @staticmethod
    def backward(ctx, *grads):
        n = ctx.n
        perms, invs = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        live = [k for k in range(n) if grads[k] is not None]
        outs = [None] * n
        if live:
            gs = [grads[k].contiguous() for k in live]
            ds = [torch.empty((gs[i].shape[0], invs[k].numel()), dtype=torch.float32, device=gs[i].device) for i, k in enumerate(live)]
            _permute_multi(gs, [invs[k] for k in live], [perms[k].numel() for k in live], ds)
            for i, k in enumerate(live):
                outs[k] = ds[i]
        ret = [None]
        for k in range(n):
            ret += [outs[k], None, None]
        return tuple(ret)


def _permute_multi(srcs, perms, cs, dsts):
    n = len(srcs)
    vp, i64 = ctypes.c_void_p * n, ctypes.c_int64 * n
    ops._run("permute_cols", srcs[0], _lib.load().mp_permute_cols_multi_f32, n, vp(*[t.data_ptr() for t in srcs]), vp(*[t.data_ptr() for t in perms]),
             i64(*[t.shape[0] for t in srcs]), i64(*cs), i64(*[p_.numel() for p_ in perms]), vp(*[t.data_ptr() for t in dsts]))


_PERMS = {}
_PREPERMUTED = {}     # id(first conv of a level) -> its weight in the internal column order, produced by prepermute() for the next forward


def prepermute(levels):
    """levels: [(first conv of a set-abstraction level, layout)] of every level the coming forward will run.  The first-layer weights of
    all of them are brought into the internal column order with ONE launch (and their gradients back with one at the end of backward);
    shared_mlp_max picks the result up instead of permuting its own."""
    todo = []
    for conv, layout in levels:
        cin = conv.in_channels
        cpad = (cin + 3) // 4 * 4
        rotate = layout == "feats_first" and cin > 3
        if (rotate or cpad != cin) and conv.weight.is_cuda:
            todo.append((conv, cin, cpad, rotate))
    _PREPERMUTED.clear()
    if len(todo) < 2:
        return
    args = []
    for conv, cin, cpad, rotate in todo:
        w = conv.weight.view(conv.out_channels, conv.in_channels)
        args += [w, *_first_weight_perm(cin, cpad, rotate, w.device)]
    outs = _PermuteColsMulti.apply(len(todo), *args)
    for (conv, *_), o in zip(todo, outs):
        _PREPERMUTED[id(conv)] = (o, conv.weight._version, weakref.ref(conv))     # (the module itself: an id() is reused once its owner is gone)


def _first_weight_perm(cin, cpad, rotate, device):
    """Column maps of the first layer's weight: reference order [xyz(3) | feats] -> internal [feats | xyz | zero pad]
    (rotate) or just the zero padding; cached per shape and device."""
    key = (cin, cpad, rotate, device)
    if key not in _PERMS:
        order = (list(range(3, cin)) + [0, 1, 2]) if rotate else list(range(cin))
        perm = order + [-1] * (cpad - cin)
        inv = [0] * cin
        for c, j in enumerate(order):
            inv[j] = c
        _PERMS[key] = (torch.tensor(perm, dtype=torch.int32).to(device), torch.tensor(inv, dtype=torch.int32).to(device))
    return _PERMS[key]


def shared_mlp_max(grouped, convs, bns, layout="xyz_first", dtype="f32", sync_bn=None, grad_to=None):
    """grouped [B,S,K,C (+ zero padding up to a multiple of 4)] -> [B,S,Cout] = max_K relu(bn(conv(.))) chained over the
    layers (fused HIP path).

    layout describes the channel order of `grouped` relative to the first conv's input channels:
      "xyz_first"   : as the module's weight expects them (reference order, models/pointnet2_utils.py:138);
      "feats_first" : [features (Cin-3), centred xyz (3), zero pad] -- what the set-abstraction modules of this package
                      produce internally.  The first weight's columns are rotated to match, and backward computes the
                      input gradient of the feature columns only (coordinates carry no gradient on this path), which
                      drops the near-empty second 128-column tile of a 131-channel input.
    The kernels work on float4 channel groups: an input with C % 4 != 0 is zero-padded here.
    dtype "bf16": the contractions run on the bf16 matrix cores (operands rounded to bf16 as they are staged, fp32 accumulation;
    stored activations, BatchNorm, pooling and all outputs stay fp32) -- mp_sa_mlp_{fwd,bwd}_bf16.
    grad_to ("feats_first" only): the feature tensor [B, S*K, Cin-3] that `grouped` was assembled from WITHOUT autograd (the caller
    detached it): its gradient comes back compact and contiguous instead of as the leading columns of grouped's."""
    if dtype not in ("f32", "bf16"):
        raise ValueError("dtype must be 'f32' or 'bf16'")
    # sync_bn: None / False = per-replica statistics; True or a process group = train-mode statistics over that group's ranks
    from .sync_bn import resolve
    sync_group = resolve(sync_bn)
    import torch.nn.functional as F
    ops._need_hip(grouped)
    B, S, K, C = grouped.shape
    cin = convs[0].in_channels
    cpad = (cin + 3) // 4 * 4
    if C == cin and cpad != cin:
        grouped = F.pad(grouped, (0, cpad - cin))
        C = cpad
    if C != cpad:
        raise ValueError(f"grouped has {C} channels, the first conv expects {cin}")
    x = grouped.reshape(B * S * K, C)
    x = x.contiguous() if x.dtype == torch.float32 else x.contiguous().float()
    training = bns[0].training
    params = []
    for i, (conv, bn) in enumerate(zip(convs, bns)):
        if bn.training != training:
            raise ValueError("all BatchNorm layers of a set-abstraction level must share one mode")
        if conv.out_channels % 4:
            raise NotImplementedError("fused set-abstraction MLP: layer widths must be multiples of 4")
        w = conv.weight.view(conv.out_channels, conv.in_channels)
        if i == 0:
            rotate = layout == "feats_first" and cin > 3
            if rotate or cpad != cin:   # one launch (and one in backward) instead of cat + pad and their autograd
                pre = _PREPERMUTED.pop(id(conv), None)      # (all levels' first weights permuted together: prepermute())
                if pre is not None and (pre[2]() is not conv or pre[1] != conv.weight._version or pre[0].shape != (conv.out_channels, cpad)):
                    pre = None                              # left over from a forward that never reached this level (or from a module that is gone)
                w = pre[0] if pre is not None else _PermuteCols.apply(w, *_first_weight_perm(cin, cpad, rotate, w.device))
        track = bn.track_running_stats and bn.running_mean is not None
        if not training and not track:
            raise NotImplementedError("eval-mode BatchNorm without running statistics")
        params += [w, conv.bias, bn.weight, bn.bias, bn.running_mean if track else None,
                   bn.running_var if track else None]
    if training:
        counters = [bn.num_batches_tracked for bn in bns if bn.track_running_stats and bn.num_batches_tracked is not None]
        if counters:
            if DEFERRED_TICKS is not None:
                DEFERRED_TICKS.extend(counters)   # a training harness advances every counter of the step with ONE launch
            else:
                torch._foreach_add_(counters, 1)  # one launch for the level instead of one per BatchNorm
    bn0 = bns[0]
    momentum = bn0.momentum if bn0.momentum is not None else 1.0 / max(float(bn0.num_batches_tracked), 1.0)
    grad_cols = cin - 3 if (layout == "feats_first" and cin > 3) else 0
    writeback = _widen_interior(params, [c.out_channels for c in convs]) if (sync_group is False and WIDEN_INTERIOR) else []
    states = [bn_state(bn, params[6 * i].shape[0]) for i, bn in enumerate(bns)] if (training and sync_group is False and BN_FUSED) else None
    out = _SharedMLPMax.apply(x, grad_to, K, training, momentum, bn0.eps, len(convs), grad_cols, dtype == "bf16", sync_group, states, *params)
    if training:
        for dst, src in writeback:          # running statistics of the real channels back into the module's buffers
            dst.copy_(src[:dst.numel()])
    return out.view(B, S, -1)


# The position-stream kernels (one pass over Z per layer, forward and backward) exist for layer widths 64 / 128 / 256; every other
# width takes the tiled GEMMs (separate dX and dW passes).  An INTERIOR width between them -- the 96 of the multi-scale level
# [64, 96, 128] at 2 M positions -- is cheaper carried as 128 with 32 dead channels: zero weight rows / bias, gamma 1, beta 0 give
# z = 0, BatchNorm maps that to exactly 0, ReLU keeps it, the next layer's zero columns ignore it, and every gradient of the dead
# channels is exactly 0 (dy = 0 there).  The padding is made of autograd ops, so the parameters keep their shapes and gradients.
WIDEN_INTERIOR = True


def _widen_interior(params, widths):
    """In place on the per-layer [w, bias, gamma, beta, running_mean, running_var] list.  Returns [(module buffer, widened copy)]."""
    import torch.nn.functional as F
    writeback = []
    for i, c in enumerate(widths[:-1]):
        if not 64 < c < 128:
            continue
        p = 128 - c
        w, bias, gamma, beta, rm, rv = params[6 * i:6 * i + 6]
        params[6 * i + 0] = F.pad(w, (0, 0, 0, p))
        params[6 * i + 1] = None if bias is None else F.pad(bias, (0, p))
        params[6 * i + 2] = F.pad(gamma, (0, p), value=1.0)
        params[6 * i + 3] = F.pad(beta, (0, p))
        if rm is not None:
            rmw, rvw = F.pad(rm.detach(), (0, p)), F.pad(rv.detach(), (0, p), value=1.0)
            params[6 * i + 4], params[6 * i + 5] = rmw, rvw
            writeback += [(rm, rmw), (rv, rvw)]
        params[6 * (i + 1)] = F.pad(params[6 * (i + 1)], (0, p))
    return writeback
