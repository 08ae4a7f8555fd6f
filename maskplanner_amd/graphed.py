"""The drop-in model's `forward` (and its backward) replayed from two hipGraphs inside an UNCHANGED training loop.

The reference's loop body (train_maskplanner.py:182-227) calls `model(point_cloud)`, `loss_handler.compute(...)`, `loss.backward()`,
`opt.step()`: launched op by op from Python the step follows the host (4.8 ms for 2.1 ms of device work, tools/dropin_phases.py).  What the
loop cannot see is HOW `model(...)` runs, so after a few eager calls with one input shape the model records

  graph F   everything `forward` launches for a batch of that shape (sampling of both levels with the FPS start indices as static inputs,
            the three set abstractions, the heads with their BatchNorm statistics and torch's own dropout: graph-safe Philox offsets), and
  graph R   the backward of F's autograd graph from static output gradients into static parameter gradients,

and from then on `forward` = copy the batch into F's input, draw the FPS starts the way the reference does (CPU generator, :77), replay F,
hand out copies of the outputs behind ONE autograd node whose backward copies the output gradients in, replays R and publishes the
parameter gradients as `param.grad` (assigned, or added when the loop accumulates).  The published tensors are the recording's own buffers:
they hold their values until the model's next forward (the optimizer step, clipping, logging in between read them in place); a loop that
keeps them past that point -- accumulation without zero_grad -- gets them copied out when that forward starts.  Same kernels, same order, same arithmetic as the eager
module code -- the graphs are recorded FROM it.

Forwards under `torch.no_grad()` (train_maskplanner.py:385-389, test_maskplanner.py:226-230) replay a graph F of their own mode (MASKPLANNER_DROPIN_GRAPH_EVAL=0:
eager).  Left alone (eager, as before): forwards with autograd on in eval mode, inputs that require a gradient,
a second forward before the first one's backward or while the first one's autograd graph is alive (both passes would share the static
buffers), a model or parameter that carries hooks (they would stop firing: `_hooked`), a model with a factor store (harness.TrainStep
schedules `encode` / `heads` itself), a live process group (DDP's hooks hang off AccumulateGrad, which this path bypasses), shapes seen
fewer than WARM times, and everything after a failed recording.  MASKPLANNER_DROPIN_GRAPH=0 switches the path off -- needed for
`torch.autograd.grad(loss, model.parameters())`: the replaying node publishes the parameter gradients as `.grad` (it is not connected to the
parameters' AccumulateGrad nodes: 150 copies per step), so a caller that asks autograd for them as return values gets "unused" inputs.
"""
import os
import warnings
import weakref

import torch

from . import pointnet2_utils as pu

ENABLED = os.environ.get("MASKPLANNER_DROPIN_GRAPH", "1") != "0"
EVAL = os.environ.get("MASKPLANNER_DROPIN_GRAPH_EVAL", "1") != "0"     # forwards under torch.no_grad() (the reference's eval / test loops): graph F alone
FORK_IN_CAPTURE = os.environ.get("MASKPLANNER_DROPIN_GRAPH_FORK", "0") != "0"     # the second level's sampling as a parallel branch of graph F
WARM = 3                 # eager calls with a shape before it is recorded (lazy initialisation, the allocator, kernel selection)
MAX_SHAPES = 2           # recorded shapes per model and mode (the full batch and an epoch's last, smaller one)


_DBG = os.environ.get("MASKPLANNER_DROPIN_GRAPH_DEBUG") == "1"


def _dbg(*a):
    if _DBG:
        print("[graphed]", *a, flush=True)


class _Replay(torch.autograd.Function):
    """One autograd node for the whole model: forward = replay F, backward = replay R."""

    @staticmethod
    def forward(ctx, runner, token, anchor):
        runner.graph_f.replay()
        ctx.runner, ctx.ticket, ctx.token = runner, runner.ticket, token        # (token: see _Token -- it dies with this node)
        return tuple(o.detach().clone() for o in runner.outs_t)

    @staticmethod
    def backward(ctx, *gouts):
        r = ctx.runner
        if ctx.ticket != r.ticket or not r.pending:
            raise RuntimeError("maskplanner_amd.graphed: backward through a forward whose recorded buffers were reused "
                               "(a second backward, or a later forward of the same shape ran first)")
        it = iter(gouts)
        for o, s in zip(r.outs_t, r.gouts):
            g = next(it)
            if s is None:
                continue
            if g is None:
                s.zero_()
            else:
                s.copy_(g)
        # a loop that accumulates over several backward passes still holds the static buffers as .grad: move the old values out first
        for p, g in r.grads:
            if p.grad is g:
                p.grad = g.clone()
        r.graph_r.replay()
        for p, g in r.grads:
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
        r.pending = False
        return None, None, None


class _Token:
    """Lives as long as the autograd node it is stored on (`ctx.token`): the recorded buffers stay reserved for that node's backward until the
    NODE is gone -- the tensors that were handed out may die long before it (`loss = f(batch)` keeps the graph, not the model's outputs)."""
    __slots__ = ("__weakref__",)


class _Runner:
    model = property(lambda self: self._model())

    def __init__(self, model, xyz, train):
        self._model, self.train = weakref.ref(model), train        # (weak: the model owns this runner -- a strong reference would make the pair a cycle,
                                                                   # freed whenever the collector runs, e.g. in the middle of somebody's recording)
        self.failed, self.calls, self.pending, self.ticket, self.live = False, 0, False, 0, []
        self.graph_f = self.graph_r = None
        self.draws, self.ptrs = [], ()
        self.shape, self.strides = tuple(xyz.shape), tuple(xyz.stride())

    def _record(self, xyz):
        from .harness import _capture_kw, recording
        model, dev = self.model, xyz.device
        B, C, N = xyz.shape
        # the loop hands over point_cloud.permute(0, 2, 1).to(device): points-major storage behind a channels-major view
        self.x = torch.empty(B, N, C, device=dev, dtype=xyz.dtype).permute(0, 2, 1) if xyz.stride(1) == 1 else torch.empty_like(xyz)
        self.x.copy_(xyz)
        params = [p for p in model.parameters() if p.requires_grad]
        self.anchor = params[0] if params else None
        _dbg("rehearse")
        self._rehearse(params)
        _dbg("rehearsed")
        kw = _capture_kw()
        if os.environ.get("MASKPLANNER_DROPIN_GRAPH_MODE"):
            kw = {"capture_error_mode": os.environ["MASKPLANNER_DROPIN_GRAPH_MODE"]}
        torch.cuda.synchronize(dev)
        gf = torch.cuda.CUDAGraph()
        # static inputs live OUTSIDE the graphs' memory pool: a tensor allocated while recording may sit in a block an earlier temporary of
        # the same recording left, and that temporary's kernel would overwrite it in every replay
        self.starts = [(torch.empty(b, dtype=torch.long, device=dev), n) for b, n in self.draws]
        pu._capture_starts = list(self.starts)
        # The recorded forward runs on ALIASES of the parameters (same storage, fresh autograd leaves).  A parameter's AccumulateGrad node
        # belongs to the stream its first forward ran on -- the loop's default stream -- and lives as long as any graph of an earlier
        # forward does (the reference's loop keeps `loss` from one iteration into the next): autograd would hand every parameter gradient
        # of the recorded backward over to that stream, i.e. pull the default stream into the recording, which ends it with a crash.  The
        # aliases get their nodes while the recording stream is current and nobody else holds them.
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        alias = {n: p.detach().requires_grad_(True) for n, p in zip(names, params)} if self.train else {}
        try:
            with recording(gf, **kw):
                outs = torch.func.functional_call(model, alias, (self.x,)) if alias else model._forward_eager(self.x)
                _dbg("forward recorded")
        finally:
            pu._capture_starts = None
            pu.clear_prefetched()
        self.outs = outs
        self.outs_t = [o for o in outs if o is not None]
        self.mask = [o is not None for o in outs]
        self.graph_f = gf
        if self.train and self.anchor is not None:
            req = [o for o in self.outs_t if o.requires_grad]
            self.gouts = [torch.zeros_like(o) if o.requires_grad else None for o in self.outs_t]
            # (torch.autograd.grad: the gradients are the recording's outputs, nothing is accumulated into .grad while recording)
            gr = torch.cuda.CUDAGraph()
            _dbg("recording backward")
            with recording(gr, pool=gf.pool(), **kw):
                got = torch.autograd.grad(req, [alias[n] for n in names], [g for g in self.gouts if g is not None], allow_unused=True)
            _dbg("backward recorded")
            self.grads = [(p, g) for p, g in zip(params, got) if g is not None]
            self.graph_r = gr
        torch.cuda.synchronize(dev)

    def _rehearse(self, params):
        """One eager forward (+ backward from EVERY output) on the static input, its side effects undone: whatever a kernel does on its first
        launch (code object load, the opt-in to large dynamic LDS) must not happen while a stream records -- and the loop's own eager steps
        may never have run the backward of an output its loss does not use."""
        model, dev = self.model, self.x.device
        buffers = [(b, b.clone()) for b in model.buffers()]
        rng_cpu, rng_dev = torch.get_rng_state(), torch.cuda.get_rng_state(dev)
        try:
            outs = [o for o in model._forward_eager(self.x) if o is not None]
            if self.train:
                req = [o for o in outs if o.requires_grad]
                torch.autograd.grad(req, params, [torch.zeros_like(o) for o in req], allow_unused=True)
            del outs
        finally:
            with torch.no_grad():
                for b, c in buffers:
                    b.copy_(c)
            torch.set_rng_state(rng_cpu)
            torch.cuda.set_rng_state(rng_dev, dev)
            pu.clear_prefetched()

    def stale(self, ptrs):
        """The model's parameters are no longer the tensors the graphs were recorded with (`p.data = ...`, `load_state_dict(assign=True)`)."""
        return self.graph_f is not None and self.ptrs != ptrs

    def __call__(self, xyz):
        if self.graph_f is None:
            self._record(xyz)
            self.ptrs = _walk(self.model)[3]
        else:
            self.x.copy_(xyz)
        B, dev = xyz.shape[0], xyz.device
        self.turn = getattr(self, "turn", 0) + 1
        for i, (t, n) in enumerate(self.starts):             # the reference's draws, in the reference's order (models/pointnet2_utils.py:77)
            if pu._fps_start_queue or pu._capture_starts is not None or pu._draw_log is not None:
                t.copy_(pu._draw_fps_start(B, n, dev), non_blocking=True)        # (an override / a recording in progress: the general path)
                continue
            # the same draw from the CPU generator, into one of two pinned buffers in turn and from there straight into the graph's static
            # input: no device allocation and no pageable copy in front of the replay (the device is idle while the host gets here)
            pins = self.__dict__.setdefault("start_pins", {})
            slot = pins.get((i, self.turn & 1))
            if slot is None:
                slot = pins[(i, self.turn & 1)] = [torch.empty(B, dtype=torch.long).pin_memory(), torch.cuda.Event(), False]
            if slot[2]:
                slot[1].synchronize()
            torch.randint(0, n, (B,), dtype=torch.long, out=slot[0])
            t.copy_(slot[0], non_blocking=True)
            slot[1].record()
            slot[2] = True
        if self.graph_r is not None:
            # The static gradient buffers share the graphs' memory pool with the forward's temporaries: they hold the last backward's values
            # until the NEXT forward replays.  A loop that still has them as .grad here is accumulating (no zero_grad): move the values out.
            for p, g in self.grads:
                if p.grad is g:
                    p.grad = g.clone()
            self.ticket += 1
            self.pending = True
            token = _Token()
            handed = _Replay.apply(self, token, self.anchor)
            self.live = [weakref.ref(token)]
            del token
            outs = iter(handed)
        else:
            self.graph_f.replay()
            outs = iter([o.detach().clone() for o in self.outs_t])
        return tuple(next(outs) if keep else None for keep in self.mask)


# ---- the module tree, flattened once ------------------------------------------------------------------------------------------------
# [r6] What the unchanged loop pays per step on the host -- with the device idle -- is mostly walks over the module tree: `model.zero_grad()`
# twice per iteration (train_maskplanner.py:183, 226), this module's own look at hooks / training flags / parameter storages.  torch walks
# with recursive generators and a memo set (~0.1 ms per walk for this model); here the tree is flattened ONCE into lists and every use
# re-validates them cheaply: a global counter that torch's registration hooks bump whenever ANY module registers a parameter or a
# submodule, and -- per entry, on use -- that the owner's dict still holds the very object (assignments that bypass registration).
_STRUCT = [0]


def _bump(*_a):
    _STRUCT[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_bump)
torch.nn.modules.module.register_module_module_registration_hook(_bump)


class _Flat:
    """The flattened tree of one model.  Not state: a copy or a pickle of the model starts without it; the root itself is not in the lists
    (the model owns this object: no cycle)."""
    __slots__ = ("version", "mods", "pars")

    def __init__(self, version=-1, mods=(), pars=()):
        self.version, self.mods, self.pars = version, mods, pars

    def __deepcopy__(self, memo):
        return _Flat()

    def __reduce__(self):
        return (_Flat, ())


def _flat(model):
    """version + submodules [(owner's _modules dict, name, module)] + parameters [(owner's _parameters dict, name, parameter)], cached on the
    model; shared parameters / modules once."""
    c = model.__dict__.get("_mp_flat")
    if c is not None and c.version == _STRUCT[0]:
        return c
    mods, pars, seen = [], [], {id(model)}
    stack = [model]
    while stack:
        m = stack.pop()
        for k, p in m._parameters.items():
            if p is not None and id(p) not in seen:
                seen.add(id(p))
                pars.append((m._parameters, k, p))
        for k, c_ in m._modules.items():
            if c_ is not None and id(c_) not in seen:
                seen.add(id(c_))
                mods.append((m._modules, k, c_))
                stack.append(c_)
    c = _Flat(_STRUCT[0], mods, pars)
    model.__dict__["_mp_flat"] = c
    return c


def _flat_valid(model):
    """The flattened tree, re-built if an entry is no longer what its owner's dict holds."""
    for _attempt in range(2):
        c = _flat(model)
        if all(d.get(k) is m for d, k, m in c.mods) and all(d.get(k) is p for d, k, p in c.pars):
            return c
        model.__dict__.pop("_mp_flat", None)
    return _flat(model)


def fast_zero_grad(model, set_to_none=True):
    """`nn.Module.zero_grad()` for the drop-in models (the reference's loop calls it twice per iteration): the same effect through the flattened
    parameter list, every entry checked against its owner's dict on the way."""
    if not set_to_none or getattr(model, "_is_replica", False):
        return torch.nn.Module.zero_grad(model, set_to_none)
    for _attempt in range(2):
        c = _flat(model)
        stale = False
        for d, k, p in c.pars:
            if d.get(k) is not p:
                stale = True
                break
            if p.grad is not None:
                p.grad = None
        if not stale and all(d.get(k) is m for d, k, m in c.mods):
            return
        model.__dict__.pop("_mp_flat", None)
    torch.nn.Module.zero_grad(model, set_to_none)


def _walk(model):
    """ONE pass per call over the flattened module tree (model.parameters() / model.modules() walk it with a memo set each: four of them cost
    a batch-of-one forward 0.06 ms of host time): (modules in training mode, parameters that require a gradient, anything hooked, the parameters'
    storages).  Hooks: forward / pre / backward hooks on a submodule fire only while the eager code runs (a replay launches kernels, not
    modules), tensor hooks and post-accumulate hooks on a parameter never fire (the replaying node assigns `.grad` itself) -- a model that
    carries one stays on the eager path."""
    import torch.nn.modules.module as M
    hooked = bool(M._global_forward_hooks or M._global_forward_pre_hooks or M._global_backward_hooks or M._global_backward_pre_hooks)
    c = _flat_valid(model)
    n_train, n_req = int(model.training), 0
    if model._forward_hooks or model._forward_pre_hooks or model._backward_hooks or model._backward_pre_hooks:
        hooked = True
    for _d, _k, m in c.mods:
        n_train += m.training
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
            hooked = True
    ptrs = []
    for _d, _k, p in c.pars:
        ptrs.append(p.data_ptr())       # (a replaced parameter storage makes the recording stale: _Runner.stale)
        if p.requires_grad:
            n_req += 1
            if p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
                hooked = True
    return n_train, n_req, hooked, tuple(ptrs)


def _hooked(model):
    return _walk(model)[2]


def _eligible(model, xyz):
    if not ENABLED or not isinstance(xyz, torch.Tensor) or not xyz.is_cuda or xyz.dim() != 3 or xyz.requires_grad:
        return None
    if torch.cuda.is_current_stream_capturing() or getattr(model, "factor_store", None) is not None:
        return None
    train = model.training and torch.is_grad_enabled()
    if not train and not (EVAL and not torch.is_grad_enabled()):
        return None
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return None
    return train


class _Runners(dict):
    """The recorded graphs of one model.  Not state: a copy or a pickle of the model starts without them."""

    def __deepcopy__(self, memo):
        return _Runners()

    def __reduce__(self):
        return (_Runners, ())


def call(model, xyz):
    """`model.forward(xyz)` through the recorded graphs when the call qualifies, else None (the caller runs the eager code)."""
    train = _eligible(model, xyz)
    if train is None:
        return None
    n_train, n_req, hooked, ptrs = _walk(model)
    if hooked:
        return None
    runners = model.__dict__.setdefault("_graph_runners", _Runners())
    key = (tuple(xyz.shape), tuple(xyz.stride()), xyz.dtype, train, model.training,
           n_req if train else 0,      # (a layer frozen or thawed later: another recording)
           n_train)                    # (a BatchNorm / dropout switched to eval on its own)
    r = runners.get(key)
    if r is None:
        if len(runners) >= 2 * MAX_SHAPES:
            return None
        r = runners[key] = _Runner(model, xyz, train)
    if r.failed:
        return None
    if r.stale(ptrs):
        reset(model)
        return None
    if r.pending and all(w() is None for w in r.live):
        r.pending = False      # that forward's autograd node is gone: nobody can backpropagate through it any more
    stats = model.__dict__.setdefault("_graph_stats", {"replayed": 0, "eager": 0, "recorded": 0})
    if r.pending:              # the previous forward of this shape may still be backpropagated: its buffers are in use, this call runs eagerly
        stats["eager"] += 1
        return None
    r.calls += 1
    if r.graph_f is None and r.calls <= WARM:
        stats["eager"] += 1
        pu._draw_log = log = []          # which FPS starts one forward of this shape draws: the recording's static inputs
        try:
            out = model._forward_eager(xyz)
        finally:
            pu._draw_log = None
        r.draws = list(log)
        return out
    try:
        recorded = r.graph_f is not None
        out = r(xyz)
        stats["replayed" if recorded else "recorded"] += 1
        return out
    except Exception as exc:
        if r.graph_f is not None and r.graph_r is not None or (r.graph_f is not None and not train):
            raise          # a recorded runner that fails at replay is a bug, not a reason to fall back silently
        r.failed = True
        r.graph_f = r.graph_r = None
        pu.clear_prefetched()
        torch.cuda.synchronize()
        warnings.warn(f"maskplanner_amd.graphed: recording the model's forward/backward failed ({type(exc).__name__}: {exc}); "
                      "this shape stays on the eager path")
        return None


def stats(model):
    """Counters of the model's qualifying forwards: replayed / eager (warm-up) / recorded, and the hit rate."""
    st = dict(model.__dict__.get("_graph_stats") or {"replayed": 0, "eager": 0, "recorded": 0})
    n = st["replayed"] + st["eager"] + st["recorded"]
    st["hit_rate"] = st["replayed"] / n if n else 0.0
    return st


def reset(model):
    """Drop the recorded graphs of a model (after its parameters were replaced, e.g. `load_state_dict(assign=True)` or `.to()`)."""
    model.__dict__.pop("_graph_runners", None)


# ---------------------------------------------------------------------------------------------------------------------------------------
# LossHandler.compute(...) of the same loop (train_maskplanner.py:212-218): graph L = every launch of the weighted terms for one set of
# argument shapes and one config, graph LB = its backward from a static scalar into static gradients of the arguments that require one
# (the model's outputs).
#
# [r6] The ground-truth arguments do NOT keep their shape from batch to batch: the maskplanner alias samples trajectories at equal spacing
# (configs/maskplanner/traj_sampling_v2.yaml:1-9: "a varying number of traj points for different samples") and the collate pads every batch to
# ITS OWN maximum (utils/dataset/paintnet_ODv1.py:738-747): `y` / `stroke_ids` are [B, max n_segments of the batch, ...], `traj_as_pc` is
# [B, max n_points of the batch, 6], a new width with nearly every shuffled batch.  What makes one recording serve them all is the padding
# contract itself: rows of -100 (ids: -1) behind a sample's last real row are what the collate writes, every kernel of the terms takes a
# sample's length from the first sentinel row (pytorch3d_chamfer.py:138-149; ops.padded_lengths), and rows behind it reach no result.  So the
# recording's static ground-truth buffers are allocated at a CAPACITY -- the batch's width rounded up to the next multiple of BUCKET rows --,
# every call writes its batch into the leading columns and the sentinel behind them, and a recording serves every batch whose widths fit
# its capacities.  Host tensors (the loop hands `stroke_ids` / `traj_as_pc` over as the collate made them, loss_handler.py:629, 838) are staged
# through a persistent pinned image of the static buffer: one asynchronous copy per argument instead of a pageable one that waits.
# The config is part of the key because the reference changes loss weights between epochs (PSACDScheduler, train_maskplanner.py:168) --
# the entries the terms actually READ (LossHandler._cfg_reads), not the whole merged config of the run.
LOSS = os.environ.get("MASKPLANNER_DROPIN_GRAPH_LOSS", "1") != "0"
MAX_LOSS_KEYS = 4        # recordings kept per handler; the least recently USED one goes first
BUCKET = int(os.environ.get("MASKPLANNER_DROPIN_GRAPH_BUCKET", "128"))       # rows; capacities are multiples of it
# the collate's sentinels (paintnet_ODv1.py:743-747: add_fake_vectors_v2 -> -100 rows, add_fake_values_v2(fake_value=-1))
PAD_SENTINEL = {"y": -100.0, "traj_as_pc": -100.0, "stroke_ids": -1.0, "stroke_ids_as_pc": -1.0}


def _capacity(w):
    return max(BUCKET, -(-int(w) // BUCKET) * BUCKET)


class _LossReplay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, token, *req):
        runner.graph_l.replay()
        ctx.runner, ctx.ticket, ctx.token = runner, runner.ticket, token
        return runner.total.detach().clone()

    @staticmethod
    def backward(ctx, g):
        r = ctx.runner
        if ctx.ticket != r.ticket or not r.pending:
            raise RuntimeError("maskplanner_amd.graphed: backward through a loss whose recorded buffers were reused "
                               "(a second backward, or a later compute() with the same shapes ran first)")
        r.gout.copy_(g)
        r.graph_lb.replay()
        r.pending = False
        return (None, None) + tuple(r.in_grads)


class _LossRunner:
    def __init__(self, caps):
        self.failed, self.calls, self.pending, self.ticket, self.live = False, 0, False, 0, None
        self.graph_l = self.graph_lb = None
        self.caps = dict(caps)           # padded argument -> capacity (rows along dim 1)

    def covers(self, widths):
        return all(self.caps.get(k, -1) >= w for k, w in widths.items()) and len(widths) == len(self.caps)

    # -- the ground-truth arguments: leading columns = the batch, sentinel behind them --------------------------------------------------
    def _stage(self, k, v):
        st, fill, w = self.static[k], PAD_SENTINEL[k], v.shape[1]
        if v.is_cuda:
            st[:, :w].copy_(v, non_blocking=True)
            if w < self.width[k]:
                st[:, w:self.width[k]].fill_(fill)
        else:
            pins = self.pins.get(k)
            if pins is None:       # two pinned images of the static buffer in turn: the copy of call i may still be read when call i + 1 fills its own
                pins = self.pins[k] = [dict(buf=torch.full(st.shape, fill, dtype=st.dtype).pin_memory(), ev=torch.cuda.Event(), busy=False, w=st.shape[1])
                                       for _ in range(2)]
            slot = pins[self.ticket % 2]
            if slot["busy"]:
                slot["ev"].synchronize()
            # (numpy views: a torch CPU copy of this size fans out over the intra-op thread pool, whose workers then spin on every core of the
            # host -- on a shared box the loop's own thread lost ~90 ms to them every few steps)
            buf = slot.get("np")
            if buf is None:
                buf = slot["np"] = slot["buf"].numpy()
            buf[:, :w] = v.detach().numpy()
            if w < slot["w"]:
                buf[:, w:slot["w"]] = fill
            slot["w"] = w
            buf = slot["buf"]
            st.copy_(buf, non_blocking=True)
            slot["ev"].record()
            slot["busy"] = True
        self.width[k] = w

    def _record(self, handler, args):
        from .harness import _capture_kw, recording
        dev = args["y_pred"].device
        self.names = [k for k, v in args.items() if isinstance(v, torch.Tensor)]
        self.static, self.req, self.pins, self.width = {}, [], {}, {}
        for k in self.names:
            v = args[k]
            if k in self.caps:
                shape = (v.shape[0], self.caps[k]) + tuple(v.shape[2:])
                self.static[k] = torch.full(shape, PAD_SENTINEL[k], dtype=v.dtype, device=dev)
                self.width[k] = self.caps[k]
                with torch.no_grad():
                    self._stage(k, v)
                continue
            st = torch.empty(v.shape, dtype=v.dtype, device=dev)
            with torch.no_grad():              # (a LEAF: copying a tensor with history into it under autograd would hang it onto that history)
                st.copy_(v)
            if v.requires_grad:
                st.requires_grad_(True)
                self.req.append(k)
            self.static[k] = st
        call = dict(args)
        call.update(self.static)
        # rehearsal (see _Runner._rehearse): every kernel of the terms and of their backward launched once outside a recording -- on leaves
        # of its own, so that the static ones meet autograd for the first time on the recording stream
        reh = dict(call)
        reh.update({k: self.static[k].detach().clone().requires_grad_(True) for k in self.req})
        total, _ = handler._terms(**reh)
        torch.autograd.grad(total, [reh[k] for k in self.req], allow_unused=True)
        del total, reh
        kw = _capture_kw()
        torch.cuda.synchronize(dev)
        gl = torch.cuda.CUDAGraph()
        with recording(gl, **kw):
            total, values = handler._terms(**call)
            from .loss_handler import _pack_values
            values = _pack_values(values, handler)          # (term values + the matching status's failure flag: ONE copy to the host per call)
        self.total, self.values, self.status = total, values, getattr(handler, "last_match_status", None)
        self.gout = torch.ones_like(total)
        glb = torch.cuda.CUDAGraph()
        with recording(glb, pool=gl.pool(), **kw):
            got = torch.autograd.grad(total, [self.static[k] for k in self.req], self.gout, allow_unused=True)
        self.in_grads = list(got)
        self.graph_l, self.graph_lb = gl, glb
        torch.cuda.synchronize(dev)

    def __call__(self, handler, args):
        self.ticket += 1
        if self.graph_l is None:
            self._record(handler, args)
        else:
            with torch.no_grad():
                for k in self.names:
                    if k in self.caps:
                        self._stage(k, args[k])
                    else:
                        self.static[k].copy_(args[k], non_blocking=True)
        handler.last_match_status = self.status
        self.pending = True
        token = _Token()
        total = _LossReplay.apply(self, token, *[args[k] for k in self.req])
        self.live = weakref.ref(token)
        del token
        return total, self.values


def _cfg_signature(handler):
    """The config entries the handler's terms have read so far, with their current values (None: nothing read yet -- the first call)."""
    reads = getattr(handler, "_cfg_reads", None)
    if not reads:
        return None
    cfg = handler._cfg(track=False)
    return tuple((k, repr(cfg.get(k))) for k in sorted(reads))


def _loss_key(handler, args):
    """(what must be equal for a recording to serve the call, {padded ground-truth argument: its width}) or None."""
    sig, widths = [], {}
    for k in sorted(args):
        v = args[k]
        if isinstance(v, torch.Tensor):
            if k in PAD_SENTINEL and v.dim() >= 2 and not v.requires_grad and v.dtype.is_floating_point:
                widths[k] = v.shape[1]
                sig.append((k, v.shape[0], tuple(v.shape[2:]), v.dtype, "padded"))
            else:
                sig.append((k, tuple(v.shape), v.dtype, v.device.type, v.requires_grad))
        elif v is None or isinstance(v, (int, float, bool, str)):
            sig.append((k, v))
        else:
            return None                    # (lists of per-sample tensors and the like: not a fixed set of buffers)
    cfg = _cfg_signature(handler)
    if cfg is None:
        return None
    return (tuple(sig), tuple(handler.loss), cfg), widths


def loss_stats(handler):
    """Counters of one handler's compute() calls that qualified for the recorded path: replayed / eager (warm-up, buffers in use) / recorded /
    evicted, the recordings' capacities, and the hit rate = replayed / (replayed + eager + recorded)."""
    st = dict(handler.__dict__.get("_graph_stats") or {"replayed": 0, "eager": 0, "recorded": 0, "evicted": 0})
    n = st["replayed"] + st["eager"] + st["recorded"]
    st["hit_rate"] = st["replayed"] / n if n else 0.0
    st["capacities"] = [dict(r.caps) for r in handler.__dict__.get("_graph_runners", {}).values() if r.graph_l is not None]
    return st


def loss_call(handler, args, return_list=True):
    """(total, stacked detached term values) through the recorded graphs when the call qualifies, else None.  Only the form the reference's loop
    uses (`return_list=True`, train_maskplanner.py:212): a harness that schedules the step itself (harness.TrainStep: return_list=False) keeps
    its launches -- its per-kernel profile and its own recordings are made of them."""
    yp = args.get("y_pred")
    if (not return_list or not ENABLED or not LOSS or not isinstance(yp, torch.Tensor) or not yp.is_cuda or not yp.requires_grad or not torch.is_grad_enabled()
            or torch.cuda.is_current_stream_capturing() or (torch.distributed.is_available() and torch.distributed.is_initialized())):
        return None
    key = _loss_key(handler, args)
    if key is None:
        return None
    fixed, widths = key
    runners = handler.__dict__.setdefault("_graph_runners", _Runners())
    stats = handler.__dict__.setdefault("_graph_stats", {"replayed": 0, "eager": 0, "recorded": 0, "evicted": 0})
    # the eager warm-up calls are counted per argument signature, not per capacity or config: what they are for (lazy initialisation, the
    # allocator, kernel selection) depends on neither, so a batch that needs a larger capacity later, or an epoch with rescheduled weights,
    # is recorded at its first sighting
    seen = handler.__dict__.setdefault("_graph_seen", {})
    warm = fixed[:2]
    seen[warm] = seen.get(warm, 0) + 1
    if len(seen) > 4 * MAX_LOSS_KEYS:
        seen.pop(next(iter(seen)))
    # a recording whose capacities hold this batch serves it (the smallest such); a shape one of them covers is never recorded again
    full, r = None, None
    for k, cand in runners.items():
        if k[0] == fixed and cand.graph_l is not None and cand.covers(widths):
            if r is None or sum(cand.caps.values()) < sum(r.caps.values()):
                full, r = k, cand
    if r is None:
        if seen[warm] <= WARM:
            stats["eager"] += 1
            return None
        caps = tuple(sorted((k, _capacity(w)) for k, w in widths.items()))
        full = (fixed, caps)
        r = runners.get(full)
        if r is None:
            while len(runners) >= MAX_LOSS_KEYS:
                runners.pop(next(iter(runners)))       # least recently used (an earlier epoch's weights, a capacity no batch asks for any more)
                stats["evicted"] += 1
            r = runners[full] = _LossRunner(caps)
    runners[full] = runners.pop(full)                  # most recently used: last
    if r.failed:
        return None
    if r.pending and (r.live is None or r.live() is None):
        r.pending = False      # that call's autograd node is gone: nobody can backpropagate through it any more
    if r.pending:
        stats["eager"] += 1
        return None
    r.calls += 1
    try:
        recorded = r.graph_l is not None
        out = r(handler, args)
        stats["replayed" if recorded else "recorded"] += 1
        return out
    except Exception as exc:
        if r.graph_l is not None:
            raise
        r.failed = True
        torch.cuda.synchronize()
        warnings.warn(f"maskplanner_amd.graphed: recording the loss failed ({type(exc).__name__}: {exc}); these shapes stay on the eager path")
        return None
