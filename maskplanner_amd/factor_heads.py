"""Rank-B gradient handling for the weight-heavy regression heads (scope table row f1 "heads + optimizer", next).

fc3 / fc_normals / sm_fc3 (models/pointnet2_cls_ssg.py:270-290) hold 97 % of MaskPlanner's parameters, and the gradient
of each is `dW = g^T x` with only B rows of factors.  With `FactorLinear` the backward pass keeps the factors instead
of forming dW, and `FactorAdam` applies torch.optim.Adam's update (train_maskplanner.py:159) with the gradient rebuilt
on the fly inside one fused kernel (csrc/adam_lowrank.hip).  Under data parallelism the factors of all ranks are
all-gathered (a few MB) instead of all-reducing 137 MB of dW.

Opt-in: the drop-in modules behave like plain nn.Linear unless a model's `factor_store` is set (the harness does).
"""
import ctypes

import torch
import torch.distributed as dist
import torch.nn.functional as F

import os

from . import _lib, dp, ops

WIDE_LINEAR = True   # False (tests): plain nn.Linear (rocBLAS backward) for the wide heads of a model without a factor store


def _linear_fwd(x, weight, bias):
    """F.linear for a skinny batch: one pass over W on the fp32 matrix cores (csrc/head_linear.hip, the block kernel without its
    BatchNorm epilogue) where it applies -- the libraries' GEMMs for 32 rows stream the 49 MB heads at 2.2 TB/s."""
    if (x.is_cuda and x.dtype == torch.float32 and x.ndim == 2 and x.is_contiguous() and weight.dtype == torch.float32
            and weight.is_contiguous() and (bias is None or bias.dtype == torch.float32)
            and _lib.load().mp_head_block_supported(x.shape[0], x.shape[1], weight.shape[0])):
        B, I = x.shape
        O = weight.shape[0]
        y = torch.empty((B, O), dtype=torch.float32, device=x.device)
        p = ops._p
        ops._run("head_linear", x, _lib.load().mp_head_block_fwd_f32, p(x), p(weight), p(bias), B, I, O, 0, 0, 0.0, 0.0, None, None, None, None,
                 None, p(y), None, None, 0.0, None, 0)
        return y
    return F.linear(x, weight, bias)


def _dx_skinny(g, weight):
    """grad_x = g W by the ordered streaming kernel (csrc/adam_lowrank.hip: linear_dx_skinny_kernel, bit-reproducible; 32 batch rows per
    pass over W -- a larger batch takes one pass per slab of 32)."""
    lib = _lib.load()
    B, O = g.shape
    I = weight.shape[1]
    gx = torch.empty((B, I), dtype=torch.float32, device=g.device)
    for b0 in range(0, B, 32):
        gs = g[b0:b0 + 32]
        n = gs.shape[0]
        ws = torch.empty((lib.mp_linear_dx_skinny_workspace_bytes(n, O, I),), dtype=torch.uint8, device=g.device)
        ops._run("linear_dx_skinny", g, lib.mp_linear_dx_skinny_f32, gs.data_ptr(), weight.data_ptr(), n, O, I, gx[b0:b0 + 32].data_ptr(),
                 ws.data_ptr(), ws.numel())
    return gx


class _FactorLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, store, key):
        ctx.save_for_backward(x, weight)
        ctx.store, ctx.key, ctx.has_bias = store, key, bias is not None
        ctx.bias = bias
        return _linear_fwd(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        ctx.store[ctx.key] = (x.detach(), g)              # the factors of dW = g^T x; dW itself is never formed
        gx = None
        if ctx.needs_input_grad[0]:
            B, O = g.shape
            I = weight.shape[1]
            # the streaming kernel pays off on the wide heads (O ~ 6000-12000 rows of W); the 1024 x 1024 layers have too few
            # row slabs to fill the chip and stay on rocBLAS
            if (g.is_cuda and B <= 64 and O >= 4096 and I % 128 == 0 and g.dtype == torch.float32 and weight.is_contiguous()
                    and not ops.DETERMINISTIC):
                # the stream of W through the matrix cores (csrc/linear_dx.hip; atomics between its K slices: not bit-reproducible)
                gx = ops.zeroed_empty((B, I), torch.float32, g.device)
                ops._run("linear_dx_mfma", g, _lib.load().mp_linear_dx_mfma_f32, g.data_ptr(), weight.data_ptr(), B, O, I, gx.data_ptr())
            elif g.is_cuda and B <= 64 and O >= 4096 and I % 4 == 0 and g.dtype == torch.float32 and weight.is_contiguous():
                gx = _dx_skinny(g, weight)
            else:
                gx = g @ weight
        gb = None
        if ctx.has_bias:
            pending = ctx.store.get(BIAS_QUEUE)
            if pending is not None and g.is_cuda:
                pending.append((ctx.bias, g))     # all bias gradients of the step in one launch: flush_bias_grads()
            else:
                gb = g.sum(0)
        return gx, None, gb, None, None


class _HeadBlock(torch.autograd.Function):
    """dropout(relu(bn(linear(x)))) of one head block (models/pointnet2_cls_ssg.py:309-327) as ONE launch forward and one backward
    (csrc/head_linear.hip): the Linear on the fp32 matrix cores with the BatchNorm1d statistics, ReLU and the counter-based dropout
    mask in its epilogue; backward = BatchNorm + ReLU + dropout backward, dgamma / dbeta and grad_x = dz W.  The weight gradient
    stays as factors in `store[key]` (or, without a store, is materialised by the rank-B outer-product kernel)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, training, momentum, eps, drop_p, rng, layer, store, key):
        B, I = x.shape
        O = weight.shape[0]
        z = torch.empty((B, O), dtype=torch.float32, device=x.device)
        y = torch.empty((B, O), dtype=torch.float32, device=x.device)
        stats = torch.empty((2, O), dtype=torch.float32, device=x.device)
        p = ops._p
        ops._run("head_block", x, _lib.load().mp_head_block_fwd_f32, p(x), p(weight), p(bias), B, I, O, 1, int(training), float(momentum),
                 float(eps), p(gamma), p(beta), p(running_mean), p(running_var), p(z), p(y), stats[0].data_ptr(), stats[1].data_ptr(),
                 float(drop_p) if rng is not None else 0.0, p(rng), int(layer))
        ctx.save_for_backward(x, weight, z, y, gamma, stats)
        ctx.bias = bias
        ctx.meta = (bool(training), float(drop_p) if rng is not None else 0.0, store, key)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, weight, z, y, gamma, stats = ctx.saved_tensors
        training, drop_p, store, key = ctx.meta
        B, I = x.shape
        O = weight.shape[0]
        lib = _lib.load()
        p = ops._p
        grad_y = grad_y.contiguous().float()
        dz = torch.empty((B, O), dtype=torch.float32, device=x.device)
        gg = torch.empty((O,), dtype=torch.float32, device=x.device) if (gamma is not None and ctx.needs_input_grad[3]) else None
        gbeta = torch.empty((O,), dtype=torch.float32, device=x.device) if ctx.needs_input_grad[4] else None
        if ops.DETERMINISTIC or O > 4096 or O % 4 or I % 64:
            # the bit-reproducible form: the rows kernel, then the library GEMM
            ops._run("bn_relu_rows_bwd", x, lib.mp_bn_relu_drop_rows_bwd_f32, p(grad_y), p(y), p(z), B, O, int(training), p(gamma),
                     stats[0].data_ptr(), stats[1].data_ptr(), p(dz), p(gg), p(gbeta), drop_p)
            gx = dz @ weight if ctx.needs_input_grad[0] else None
        else:
            gx = (ops.zeroed_empty((B, I), torch.float32, x.device) if lib.mp_head_block_bwd_slices(O) > 1
                  else torch.empty((B, I), dtype=torch.float32, device=x.device))
            ops._run("head_block_bwd", x, lib.mp_head_block_bwd_f32, p(grad_y), p(y), p(z), p(weight), B, I, O, int(training), p(gamma),
                     stats[0].data_ptr(), stats[1].data_ptr(), drop_p, p(dz), p(gg), p(gbeta), p(gx))
        gw = gb = None
        if store is not None:
            store[key] = (x.detach(), dz)                  # the factors of dW = dz^T x
        elif ctx.needs_input_grad[1]:
            if I % 4 == 0:
                gw = torch.empty_like(weight)
                ops._run("linear_dw_outer", dz, lib.mp_linear_dw_outer_f32, p(dz), p(x), B, O, I, p(gw))
            else:
                gw = dz.t() @ x
        if ctx.bias is not None and ctx.needs_input_grad[2]:
            pending = store.get(BIAS_QUEUE) if store is not None else None
            if pending is not None:
                pending.append((ctx.bias, dz))
            else:
                gb = dz.sum(0)
        return (gx, gw, gb, gg, gbeta) + (None,) * 10


def _bn_args(bn):
    training = bn.training or bn.running_mean is None
    track = bn.track_running_stats and bn.running_mean is not None
    momentum = bn.momentum if bn.momentum is not None else 1.0 / max(float(bn.num_batches_tracked), 1.0)
    return training, momentum, (bn.running_mean if track else None), (bn.running_var if track else None)


class _HeadBlocks2(torch.autograd.Function):
    """Two head blocks in one launch each way (mp_head_blocks_{fwd,bwd}_f32): the blocks of the two head branches that stand side by side
    in the network -- fc1 / sm_fc1 (both read the global feature: xb None, ONE grad_x, no fan-out add) and fc2 / sm_fc2 (each its own
    input).  Otherwise as _HeadBlock."""

    @staticmethod
    def forward(ctx, xa, xb, wa, ba, ga, bea, rma, rva, wb, bb, gb, beb, rmb, rvb, meta):
        shared = xb is None
        B, I = xa.shape
        dev = xa.device
        blocks = (_lib.HeadBlock * 2)()
        outs = []
        p = ops._p
        rng = meta["rng"]
        for k, (x, w, b, g, be, rm, rv) in enumerate(((xa, wa, ba, ga, bea, rma, rva), (xa if shared else xb, wb, bb, gb, beb, rmb, rvb))):
            O = w.shape[0]
            z = torch.empty((B, O), dtype=torch.float32, device=dev)
            y = torch.empty((B, O), dtype=torch.float32, device=dev)
            stats = torch.empty((2, O), dtype=torch.float32, device=dev)
            training, momentum, eps, layer = meta["bn"][k]
            blocks[k] = _lib.HeadBlock(p(x), p(w), p(b), O, 1, int(training), float(momentum), float(eps), p(g), p(be), p(rm), p(rv), p(z), p(y),
                                       stats[0].data_ptr(), stats[1].data_ptr(), float(meta["drop_p"]) if rng is not None else 0.0, p(rng), int(layer),
                                       None, None, None, None, None)
            outs.append((z, y, stats))
        ops._run("head_blocks", xa, _lib.load().mp_head_blocks_fwd_f32, 2, blocks, B, I)
        (za, ya, sa), (zb, yb, sb) = outs
        ctx.save_for_backward(xa, xb, wa, wb, za, ya, sa, zb, yb, sb, ga, gb)
        ctx.biases = (ba, bb)
        ctx.meta = meta
        return ya, yb

    @staticmethod
    def backward(ctx, gya, gyb):
        xa, xb, wa, wb, za, ya, sa, zb, yb, sb, ga, gb = ctx.saved_tensors
        meta = ctx.meta
        shared = xb is None
        B, I = xa.shape
        dev = xa.device
        lib = _lib.load()
        p = ops._p
        store = meta["store"]
        drop_p = float(meta["drop_p"]) if meta["rng"] is not None else 0.0
        blocks = (_lib.HeadBlock * 2)()
        gx_a = ops.zeroed_empty((B, I), torch.float32, dev)
        gx_b = gx_a if shared else ops.zeroed_empty((B, I), torch.float32, dev)
        res = []
        for k, (x, w, z, y, st, g, gy, gx) in enumerate(((xa, wa, za, ya, sa, ga, gya, gx_a), (xa if shared else xb, wb, zb, yb, sb, gb, gyb, gx_b))):
            O = w.shape[0]
            gy = gy.contiguous().float()
            dz = torch.empty((B, O), dtype=torch.float32, device=dev)
            gg = torch.empty((O,), dtype=torch.float32, device=dev) if g is not None else None
            gbeta = torch.empty((O,), dtype=torch.float32, device=dev)
            training = meta["bn"][k][0]
            blocks[k] = _lib.HeadBlock(None, p(w), None, O, 1, int(training), 0.0, 0.0, p(g), None, None, None, p(z), p(y), st[0].data_ptr(),
                                       st[1].data_ptr(), drop_p, None, 0, p(gy), p(dz), p(gg), p(gbeta), p(gx))
            res.append((x, w, dz, gg, gbeta, gy))
        ops._run("head_blocks_bwd", xa, lib.mp_head_blocks_bwd_f32, 2, blocks, B, I)
        out = []
        pending = store.get(BIAS_QUEUE) if store is not None else None
        for k, (x, w, dz, gg, gbeta, _gy) in enumerate(res):
            gw = gb_ = None
            key = meta["keys"][k]
            if store is not None and key is not None:
                store[key] = (x.detach(), dz)
            else:
                gw = torch.empty_like(w)
                ops._run("linear_dw_outer", dz, lib.mp_linear_dw_outer_f32, p(dz), p(x), B, w.shape[0], I, p(gw))
            if ctx.biases[k] is not None:
                if pending is not None:
                    pending.append((ctx.biases[k], dz))
                else:
                    gb_ = dz.sum(0)
            out.append((gw, gb_, gg, gbeta))
        (gwa, gba, gga, gbea), (gwb, gbb, ggb, gbeb) = out
        return (gx_a, None if shared else gx_b, gwa, gba, gga, gbea, None, None, gwb, gbb, ggb, gbeb, None, None, None)


def head_blocks2_ok(xa, xb, lin_a, bn_a, lin_b, bn_b):
    """The two blocks can share launches: each qualifies alone (head_block_ok), same input width, widths the backward kernel tiles."""
    if ops.DETERMINISTIC or not head_block_ok(xa, lin_a, bn_a) or not head_block_ok(xa if xb is None else xb, lin_b, bn_b):
        return False
    wa, wb = lin_a.weight, lin_b.weight
    return (wa.shape[1] == wb.shape[1] and wa.shape[1] % 64 == 0 and all(w.shape[0] % 4 == 0 and w.shape[0] <= 4096 for w in (wa, wb))
            and (xb is None or xb.shape == xa.shape))


def head_blocks2(xa, xb, lin_a, bn_a, lin_b, bn_b, store, key_a, key_b, drop=None, layers=(0, 0)):
    """(relu(bn_a(lin_a(xa))), relu(bn_b(lin_b(xb)))) -- xb None: both read xa -- in one launch each way; drop = (p, rng): the nn.Dropout
    behind both blocks in the same launch (mask layers `layers`).  Check head_blocks2_ok first."""
    ta, ma, rma, rva = _bn_args(bn_a)
    tb, mb, rmb, rvb = _bn_args(bn_b)
    p, rng = drop if drop is not None else (0.0, None)
    meta = dict(bn=((ta, ma, bn_a.eps, layers[0]), (tb, mb, bn_b.eps, layers[1])), drop_p=p, rng=rng, store=store, keys=(key_a, key_b))
    ya, yb = _HeadBlocks2.apply(xa, xb, lin_a.weight, lin_a.bias, bn_a.weight, bn_a.bias, rma, rva,
                                lin_b.weight, lin_b.bias, bn_b.weight, bn_b.bias, rmb, rvb, meta)
    if ops.RELU_TAP is not None:
        ops.RELU_TAP.append(ya.detach() > 0)
        ops.RELU_TAP.append(yb.detach() > 0)
    return ya, yb


def head_block_ok(x, linear, bn):
    """The one-launch block applies: fp32 [B <= 64, I] on the GPU, contiguous fp32 parameters, a width csrc/head_linear.hip tiles, and
    BatchNorm statistics local to this process."""
    w = linear.weight
    sync = getattr(bn, "sync_bn", None)
    return (x.is_cuda and x.dtype == torch.float32 and x.ndim == 2 and x.is_contiguous() and w.dtype == torch.float32 and w.is_contiguous()
            and (sync is None or sync is False) and (bn.training or bn.running_mean is not None)
            and bool(_lib.load().mp_head_block_supported(x.shape[0], x.shape[1], w.shape[0])))


def head_block(x, linear, bn, store, key, dropout=None):
    """relu(bn(linear(x))) -- with dropout = (p, rng, layer) also the nn.Dropout(p) behind it, see ops.bn_relu_rows -- in one launch;
    the caller has checked head_block_ok and advanced bn.num_batches_tracked."""
    training = bn.training or bn.running_mean is None
    track = bn.track_running_stats and bn.running_mean is not None
    momentum = bn.momentum if bn.momentum is not None else 1.0 / max(float(bn.num_batches_tracked), 1.0)
    p, rng, layer = dropout if dropout is not None else (0.0, None, 0)
    y = _HeadBlock.apply(x, linear.weight, linear.bias, bn.weight, bn.bias, bn.running_mean if track else None,
                         bn.running_var if track else None, training, momentum, bn.eps, p, rng, layer, store, key)
    if ops.RELU_TAP is not None:
        ops.RELU_TAP.append(y.detach() > 0)
    return y


class _FactorLinear2(torch.autograd.Function):
    """Two Linears fed by the same activation (fc3 / fc_normals, models/pointnet2_cls_ssg.py:311, :327; sm_fc3 / the mask-confidence
    layer, :336-338): one launch forward, one for grad_x = g1 W1 + g2 W2 (csrc/head_linear.hip, linear_dx.hip) -- no fan-out add.  A
    weight with a key keeps its gradient as factors in `store`; one without (a small dense layer) gets dW from the rank-B outer-product
    kernel."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, store, key1, key2):
        B, I = x.shape
        O1, O2 = w1.shape[0], w2.shape[0]
        y1 = torch.empty((B, O1), dtype=torch.float32, device=x.device)
        y2 = torch.empty((B, O2), dtype=torch.float32, device=x.device)
        p = ops._p
        ops._run("head_linear2", x, _lib.load().mp_head_linear2_fwd_f32, p(x), B, I, p(w1), p(b1), O1, p(y1), p(w2), p(b2), O2, p(y2))
        ctx.save_for_backward(x, w1, w2)
        ctx.biases = (b1, b2)
        ctx.meta = (store, key1, key2)
        ctx.set_materialize_grads(False)
        return y1, y2

    @staticmethod
    def backward(ctx, g1, g2):
        x, w1, w2 = ctx.saved_tensors
        store, key1, key2 = ctx.meta
        B, I = x.shape
        lib = _lib.load()
        p = ops._p
        gs = [torch.zeros((B, w.shape[0]), dtype=torch.float32, device=x.device) if g is None else g.contiguous().float()
              for g, w in ((g1, w1), (g2, w2))]
        gx = None
        if ctx.needs_input_grad[0]:
            gx = ops.zeroed_empty((B, I), torch.float32, x.device)
            ops._run("linear_dx_mfma", x, lib.mp_linear_dx_mfma2_f32, p(gs[0]), p(w1), w1.shape[0], p(gs[1]), p(w2), w2.shape[0], B, I, p(gx))
        gw, gb = [None, None], [None, None]
        pending = store.get(BIAS_QUEUE)
        for k, (w, b, g, key) in enumerate(zip((w1, w2), ctx.biases, gs, (key1, key2))):
            if key is not None:
                store[key] = (x.detach(), g)
            elif ctx.needs_input_grad[1 + 2 * k]:
                gw[k] = torch.empty_like(w)
                ops._run("linear_dw_outer", g, lib.mp_linear_dw_outer_f32, p(g), p(x), B, w.shape[0], I, p(gw[k]))
            if b is not None:
                if pending is not None:
                    pending.append((b, g))
                else:
                    gb[k] = g.sum(0)
        return gx, gw[0], gb[0], gw[1], gb[1], None, None, None


def factor_linear2(x, lin1, lin2, store, key1, key2):
    """(lin1(x), lin2(x)) for two Linears on the same input: one launch each way where the one-pass kernels apply (key None: that
    layer's weight gradient is dense), else two factor_linear calls / plain calls."""
    w1, w2 = lin1.weight, lin2.weight
    if (store is not None and not ops.DETERMINISTIC and x.is_cuda and x.dtype == torch.float32 and x.ndim == 2 and x.is_contiguous()
            and w1.dtype == torch.float32 and w2.dtype == torch.float32 and w1.is_contiguous() and w2.is_contiguous()
            and x.shape[1] % 128 == 0 and _lib.load().mp_head_block_supported(x.shape[0], x.shape[1], w1.shape[0])
            and _lib.load().mp_head_block_supported(x.shape[0], x.shape[1], w2.shape[0])):
        return _FactorLinear2.apply(x, w1, lin1.bias, w2, lin2.bias, store, key1, key2)
    return tuple(factor_linear(x, lin, store, key) if key is not None else lin(x) for lin, key in ((lin1, key1), (lin2, key2)))


BIAS_QUEUE = "__bias_grads__"     # store[BIAS_QUEUE] = []: FactorLinear queues (bias, dy) pairs instead of reducing each one


@torch.no_grad()
def flush_bias_grads(store):
    """bias.grad = dy.sum(0) for every queued head Linear, one launch (csrc/adam_multi.hip: mp_colsum_multi_f32).  Call after
    backward() and before the gradients are exchanged / consumed."""
    queue = store.get(BIAS_QUEUE)
    if not queue:
        return
    n = len(queue)
    rows = queue[0][1].shape[0]
    outs = [torch.empty_like(b) for b, _ in queue]
    arr = ctypes.c_void_p * n
    gp = arr(*[g.data_ptr() for _, g in queue])
    op = arr(*[o.data_ptr() for o in outs])
    cols = (ctypes.c_int64 * n)(*[g.shape[1] for _, g in queue])
    ops._run("colsum_multi", outs[0], _lib.load().mp_colsum_multi_f32, n, gp, op, cols, rows)
    for (b, _), o in zip(queue, outs):
        b.grad = o if b.grad is None else b.grad + o
    del queue[:]


class _WideLinear(torch.autograd.Function):
    """nn.Linear for the weight-heavy heads of a model WITHOUT a factor store (the drop-in path: torch.optim.Adam on dense
    gradients): forward = F.linear; backward on the library's streaming kernels instead of the GEMMs rocBLAS selects for a 32-row
    batch -- grad_x through csrc/linear_dx.hip (107 -> 21 us for the 11988 x 1024 heads), grad_W as a rank-B outer-product stream
    (85 -> ~15 us), grad_b a column sum."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _linear_fwd(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous().float()      # (the kernels read fp32: an autocast / half gradient is widened here, never reinterpreted)
        B, O = g.shape
        I = weight.shape[1]
        lib = _lib.load()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if ops.DETERMINISTIC:
                gx = _dx_skinny(g, weight)
            else:
                gx = torch.empty((B, I), dtype=torch.float32, device=g.device)
                ops._run("linear_dx_mfma", g, lib.mp_linear_dx_mfma_f32, g.data_ptr(), weight.data_ptr(), B, O, I, gx.data_ptr())
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(weight)
            xc = x.contiguous().float()
            ops._run("linear_dw_outer", g, lib.mp_linear_dw_outer_f32, g.data_ptr(), xc.data_ptr(), B, O, I, gw.data_ptr())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gx, gw, gb


def _wide_ok(x, linear):
    w = linear.weight
    return (WIDE_LINEAR and x.is_cuda and x.dtype == torch.float32 and x.ndim == 2 and x.shape[0] <= 64 and w.shape[0] >= 4096
            and w.shape[1] % 128 == 0 and w.dtype == torch.float32 and w.is_contiguous() and torch.is_grad_enabled())


def factor_linear(x, linear, store, key):
    """y = linear(x); if `store` is a dict the weight gradient is left as factors in store[key]; without one the wide heads
    still take the library's backward kernels (_WideLinear), everything else is plain nn.Linear."""
    if store is None:
        return _WideLinear.apply(x, linear.weight, linear.bias) if _wide_ok(x, linear) else linear(x)
    return _FactorLinear.apply(x, linear.weight, linear.bias, store, key)


class FactorAdam:
    """Adam (torch defaults: betas (0.9, 0.999), eps 1e-8, no weight decay / amsgrad) for weights whose gradient is kept
    as factors.  step() consumes `store[key] = (x [B,I], g [B,O])` for every registered weight."""

    def __init__(self, named_weights, store, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, process_group=None, capturable=False):
        self.weights = dict(named_weights)       # key -> nn.Parameter [O, I]
        self.store = store
        self.lr, self.betas, self.eps = lr, betas, eps
        self.group = process_group
        # capturable: the update count lives in ONE device scalar shared by all weights (they always step together), advanced
        # by a device op, so that a step recorded into a hipGraph keeps counting when it is replayed
        self.capturable = bool(capturable)
        self.step_dev = None
        if self.capturable and self.weights:
            dev = next(iter(self.weights.values())).device
            self.step_dev = torch.zeros((), dtype=torch.float32, device=dev)
        self.state = {k: dict(step=0, exp_avg=torch.zeros_like(w), exp_avg_sq=torch.zeros_like(w))
                      for k, w in self.weights.items()}

    def _gather(self, tensors):
        """All-gather a list of [B, n_i] factor tensors with ONE collective: returns the list with world*B rows."""
        world = dist.get_world_size(self.group)
        B = tensors[0].shape[0]
        widths = [t.shape[1] for t in tensors]
        flat = torch.cat([t.reshape(B, -1) for t in tensors], dim=1).contiguous()        # [B, sum n_i]
        out = torch.empty((world,) + tuple(flat.shape), dtype=flat.dtype, device=flat.device)
        dist.all_gather_into_tensor(out.view(world * B, -1), flat, group=self.group)
        out = out.view(world * B, -1)
        return [c.contiguous() for c in out.split(widths, dim=1)]

    @torch.no_grad()
    def step(self):
        keys = [k for k in self.weights if k in self.store]
        if not keys:
            return
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        xs = [self.store[k][0] for k in keys]
        gs = [self.store[k][1] for k in keys]
        if dp.exchanging(self.group):
            # distinct input tensors only (fc3 and fc_normals share theirs)
            uniq = {}
            for x in xs:
                uniq.setdefault(x.data_ptr(), x)
            gathered = self._gather(list(uniq.values()) + gs)
            xmap = dict(zip(uniq.keys(), gathered[:len(uniq)]))
            xs = [xmap[x.data_ptr()] for x in xs]
            gs = gathered[len(uniq):]
        lib = _lib.load()
        if self.step_dev is not None:
            self.step_dev.add_(1.0)
        for k, x, g in zip(keys, xs, gs):
            w, st = self.weights[k], self.state[k]
            st["step"] += 1
            O, I = w.shape
            ops._run("adam_lowrank", w, lib.mp_adam_lowrank_f32, w.data_ptr(), st["exp_avg"].data_ptr(),
                     st["exp_avg_sq"].data_ptr(), x.data_ptr(), g.data_ptr(), x.shape[0], O, I, 1.0 / world, self.lr,
                     self.betas[0], self.betas[1], self.eps, st["step"], None if self.step_dev is None else self.step_dev.data_ptr())
            del self.store[k]


class DenseAdam:
    """torch.optim.Adam (defaults: betas (0.9, 0.999), eps 1e-8, no weight decay / amsgrad; train_maskplanner.py:159) for the
    parameters that keep a dense gradient -- the encoder, BatchNorm affine parameters, biases: ~150 tensors, 0.9 M elements --
    as csrc/adam_multi.hip: up to 48 tensors per launch with their pointers in the kernel arguments, instead of torch's
    multi_tensor_apply with per-tensor device step counters (60-90 us per step when capturable).  `state[p]` holds `exp_avg`
    and `exp_avg_sq` like torch's optimizer state.  capturable: the step count is one device float advanced by a device op."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        self.params = [p for p in params]
        for p in self.params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise ValueError("DenseAdam drives contiguous float32 parameters on the GPU")
        self.lr, self.betas, self.eps = lr, betas, eps
        self.steps = 0
        self.step_dev = torch.zeros((), dtype=torch.float32, device=self.params[0].device) if (capturable and self.params) else None
        self.state = {p: dict(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p)) for p in self.params}
        n = len(self.params)
        arr = ctypes.c_void_p * n
        self._p = arr(*[p.data_ptr() for p in self.params])
        self._m = arr(*[self.state[p]["exp_avg"].data_ptr() for p in self.params])
        self._v = arr(*[self.state[p]["exp_avg_sq"].data_ptr() for p in self.params])
        self._g = arr()
        self._n = (ctypes.c_int64 * n)()
        self._numel = [p.numel() for p in self.params]

    @torch.no_grad()
    def step(self, ticked=False):
        """ticked: the device-side update count was already advanced for this step (the harness does it in its arena launch)."""
        if not self.params:
            return
        live = 0
        for i, p in enumerate(self.params):     # a parameter without a gradient this step is skipped (length 0), like torch does
            g = p.grad
            if g is None:
                self._n[i] = 0
                self._g[i] = None
                continue
            if not (g.is_contiguous() and g.dtype == torch.float32):
                g = p.grad = g.contiguous().float()
            self._g[i] = g.data_ptr()
            self._n[i] = self._numel[i]
            live += 1
        if not live:
            return
        self.steps += 1
        if self.step_dev is not None and not ticked:
            self.step_dev.add_(1.0)
        ops._run("adam_multi", self.params[0], _lib.load().mp_adam_multi_f32, len(self.params), self._p, self._g, self._m, self._v, self._n,
                 1.0, self.lr, self.betas[0], self.betas[1], self.eps, 0 if self.step_dev is not None else self.steps,
                 None if self.step_dev is None else self.step_dev.data_ptr())

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
