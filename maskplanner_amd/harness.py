"""Synthetic-data training harness around the same call sequence as the reference's hot loop
(train_maskplanner.py:207-221): model(point_cloud) -> LossHandler.compute(...) -> backward -> Adam step.
Used by bench.py, __graft_entry__.smoke() and the full-size tests; the reference's dataset is not public, so
batches come from maskplanner_amd.synthetic with the collated-tensor contract of the reference.
"""
import os

import numpy as np

import torch

from . import dp, synthetic
from .factor_heads import BIAS_QUEUE, DenseAdam, FactorAdam, flush_bias_grads
from . import pointnet2_utils as pu
from .loss_handler import LossHandler, maskplanner_loss_config
from .pointnet2_cls_ssg import maskplanner_model


def _capture_kw():
    """How the step is recorded while a process group is alive.  [r5] RCCL's watchdog thread polls the completion events of the
    collectives issued so far (hipEventQuery, every 100 ms, until it has reaped them); under the default `global` capture mode such a call
    from ANY thread while this thread records is an error -- it invalidates the capture AND raises inside the watchdog, which terminates the
    process (seen with one forced RCCL rank, tools/dp_overhead.py: "operation not permitted when stream is capturing").  The eager steps in
    front of the recording have issued collectives, so: record in `thread_local` mode (only this thread's calls are checked; it is the only
    one that enqueues work) after the device has drained and the watchdog has had time to reap what completed."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return {}
    torch.cuda.synchronize()
    if dist.get_backend() == "nccl":
        import time
        time.sleep(0.25)
    return {"capture_error_mode": "thread_local"}


class recording:
    """`torch.cuda.graph(...)` with the garbage collector out of the way.  [r6] torch 2.10 no longer collects before a capture, and a dead
    reference cycle that owns a pinned host buffer (or anything whose release queries an event) may then be collected WHILE a stream records:
    the release's `hipEventQuery` is illegal under capture and aborts the process from a destructor (seen in a test session: `Fatal Python
    error: Aborted`, `Garbage-collecting` inside `_record_split`).  So: collect before, no collection during."""

    def __init__(self, g, **kw):
        self.ctx = torch.cuda.graph(g, **kw)

    def __enter__(self):
        import gc
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()
        try:
            return self.ctx.__enter__()
        except BaseException:
            if self.was:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc
        try:
            return self.ctx.__exit__(*exc)
        finally:
            if self.was:
                gc.enable()


def _even(n):
    """int64 units rounded up to a 16-byte multiple."""
    return (int(n) + 1) // 2 * 2


class TrainStep:
    def __init__(self, category="cuboids", B=32, N=5120, device="cuda", seed=1235, hidden_size=(1024, 1024), lr=1e-3,
                 dist_points="cuboid", rank=0, loss_overrides=None, prefetch_sampling=False, factor_heads=True, graph=None,
                 overlap_sampling=None, encoder="ssg", mlp_dtype="f32", sync_bn=False, stream_batches=0):
        self.cat = synthetic.CATEGORIES[category] if isinstance(category, str) else category
        self.device = torch.device(device)
        torch.manual_seed(seed)  # identical initial weights on every rank
        # encoder "msg" + mlp_dtype "bf16": BASELINE configs[4] (containers, N = 10240, multi-radius grouping, bf16 matrix cores)
        self.encoder, self.mlp_dtype = encoder, mlp_dtype
        self.SAMPLING_BEHIND_A = os.environ.get("MASKPLANNER_SAMPLING_BEHIND_A", "1" if encoder == "msg" else "0") == "1"
        self.model = maskplanner_model(self.cat, hidden_size=hidden_size, encoder=encoder, mlp_dtype=mlp_dtype).to(self.device).train()
        # SyncBN (opt-in): train-mode BatchNorm statistics over the global batch, so that a data-parallel run reproduces the
        # single-device run on the concatenated batch; its per-layer collectives rule out graph replay
        self.sync_bn = bool(sync_bn) and dp.exchanging()
        if self.sync_bn:
            from . import sync_bn as _sbn
            _sbn.enable(self.model)
        self.cfg = maskplanner_loss_config(**(loss_overrides or {}))
        self.loss_handler = LossHandler(["asymm_v6_chamfer_with_stroke_masks"], self.cfg)
        fused = self.device.type == "cuda"
        # hipGraph replay of the whole step (single-process runs): the step is ~190 launches and the Python side needs
        # ~3.5 ms to enqueue them against ~4.6 ms of device time -- on a busy host the enqueue becomes the bottleneck.
        # The step is recorded once after a few eager steps and replayed; any failure to record falls back to eager.
        if graph is None:
            graph = os.environ.get("MASKPLANNER_GRAPH", "1") != "0"
        # data-parallel runs (default; MASKPLANNER_DP_GRAPH=0 launches kernel by kernel): the two graphs are recorded WITHOUT
        # collectives and optimizers; the bucketed all-reduce, DenseAdam and the factor all-gather + Adam are launched eagerly after
        # graph B (~20 launches per step instead of ~170, so eight ranks sharing one host do not become host-bound).  Replayed
        # with two ranks on one GPU (tests/test_gpu_dp.py) and with one RCCL rank (tools/rccl_single_rank.py).
        self.dp_graph = dp.exchanging() and os.environ.get("MASKPLANNER_DP_GRAPH", "1") != "0"
        # Replica guard of that path (it has run with gloo ranks on one GPU and with a single RCCL rank only -- the first real multi-GPU
        # run is the driver's): after each of the first MASKPLANNER_DP_GUARD_STEPS replayed steps every rank compares a checksum of
        # ALL its weights with the other ranks' (one small all-gather + one host read); replicas that differ or went non-finite make
        # every rank drop the graphs, take rank 0's weights and optimizer state, and go on kernel by kernel (_dp_guard).
        self._guard_left = int(os.environ.get("MASKPLANNER_DP_GUARD_STEPS", "2")) if self.dp_graph else 0
        self.dp_fell_back = False
        # [r4] SyncBN no longer rules out replay: its per-layer all-reduces (launched through the library's hook, sync_bn.Exchange) are
        # recorded into the graphs like any other node when the backend is RCCL ("nccl": captures; measured with one forced rank:
        # 3.42 -> 2.35 ms per step).  A backend that cannot be captured (gloo: host-side collectives) keeps eager launches.
        # [r5, ADVICE r4] opt-in (MASKPLANNER_SYNCBN_GRAPH=1) until a run with >= 2 real RCCL ranks has confirmed loss parity with the eager SyncBN
        # step: the only RCCL evidence so far is one forced rank, and a capture-time failure on ONE rank of many is not covered by the launch-mode vote
        sync_graph = (self.sync_bn and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
                      and os.environ.get("MASKPLANNER_SYNCBN_GRAPH", "0") == "1")
        self.use_graph = bool(graph) and fused and (not dp.exchanging() or self.dp_graph) and not prefetch_sampling and (not self.sync_bn or sync_graph)
        self._graph, self._graph_loss, self._eager_steps, self._side = None, None, 0, None
        # Pipelined first-level sampling (see _eager_step): FPS can occupy only one workgroup per cloud -- 32 of 256 CUs for
        # ~0.35 ms at B=32 -- and depends on nothing but the input cloud, so the step computes the NEXT batch's FPS + ball
        # query on a second stream underneath its own work (_launch_sampling / _hand_over).
        if overlap_sampling is None:
            overlap_sampling = os.environ.get("MASKPLANNER_OVERLAP_SAMPLING", "1") != "0"
        # ([r4] the plan of a multi-scale level holds one ball query per radius: both encoders sample on the side stream)
        self.overlap = bool(overlap_sampling) and fused and not prefetch_sampling
        self._plan_next, self._plan_cur, self._plan_stream, self._plan_ev = None, None, None, None
        # Deferred head optimizer (see _record_split): the step is recorded as TWO graphs, encoder forward | everything else, and
        # the factor Adam of the seven head matrices (0.97 GB of HBM traffic, bandwidth-bound) is launched eagerly on its own
        # stream after the second graph: it then runs underneath the NEXT step's encoder forward (MFMA-bound), which does not
        # touch the head weights; the next step's second graph waits for it.
        self._graph_b, self._graph_b2, self._factor_args, self._adam_stream, self._adam_ev = None, None, None, None, None
        self._split_adam_wanted = os.environ.get("MASKPLANNER_SPLIT_ADAM", "1") != "0"
        self._unit = torch.ones((), dtype=torch.float32, device=self.device)
        from . import ops as _ops
        self._zero_arena = _ops.ZeroArena(self.device) if self.device.type == "cuda" else None
        self._dense_ticked = False
        # train-mode dropout of the head blocks inside their BatchNorm + ReLU launches (pointnet2_cls_ssg._block): a device (seed, step)
        # pair, the step advanced once per training step.
        self._drop_rng = None
        if fused and hasattr(self.model, "heads"):
            # (the rank is mixed in: the reference's single process draws an independent mask per row of the GLOBAL batch, so
            # replicas must not share theirs)
            self._drop_rng = torch.tensor([int(seed) * 0x9E3779B1 + 12345 + int(rank) * 0x632BE5AB, 0], dtype=torch.int64, device=self.device)
            self._drop_step = self._drop_rng[1:2]
            self.model.fused_dropout = self._drop_rng
        self.factor_opt = None
        dense = list(self.model.parameters())
        if factor_heads and fused:
            # the head matrices (99 % of the parameters, all fed by a [B, 1024] feature): gradient kept as rank-B factors, Adam
            # fused with its reconstruction (factor_heads.py); everything else goes through autograd + torch.optim.Adam
            self.model.factor_store = {}
            big = {n: p for n, p in self.model.named_parameters()
                   if n in ("fc1.weight", "fc2.weight", "fc3.weight", "fc_normals.weight", "sm_fc1.weight", "sm_fc2.weight", "sm_fc3.weight")}
            self.factor_opt = FactorAdam(big, self.model.factor_store, lr=lr, capturable=self.use_graph)
            dense = [p for p in dense if all(p is not q for q in big.values())]
        self.reducer = dp.BucketedGradAllReduce(dense)
        self.reducer.deferred = self.dp_graph and self.use_graph
        if self.factor_opt is not None:
            self._reset_factor_store()
        self._static_grads = None
        self._dp_recorded = False      # the data-parallel exchange is part of the recorded graphs (_record_split)
        # train_maskplanner.py:159.  On the GPU the dense parameters go through csrc/adam_multi.hip (same update, ~150 tensors in four
        # launches).
        if fused:
            self.opt = DenseAdam(dense, lr=lr, capturable=self.use_graph)
        else:
            self.opt = torch.optim.Adam(dense, lr=lr, fused=fused, capturable=self.use_graph)
        b = synthetic.make_batch(seed + 1000 * rank, B, N, self.cat.name, dist_points)  # a different shard per rank
        self.batch = {k: (v.to(self.device) if torch.is_tensor(v) else [t.to(self.device) for t in v])
                      for k, v in b.items()}
        # Streamed inputs (stream_batches = K > 0): K different host batches of ragged dataset items rotate through the step; batch
        # k+1 is collated onto the device (maskplanner_amd.collate: one flat copy per key + the pad kernel) and its sampling plan
        # computed on the second stream WHILE step k runs (_launch_sampling).  The step's own tensors keep fixed shapes (ground
        # truth padded to the category's maximum, per-sample lengths come from the -100 sentinel anyway), so the recorded graphs
        # stay valid.  0: the same resident batch every step (the headline bench: inputs resident in HBM).
        self._stream = None
        if stream_batches and fused and not prefetch_sampling and encoder == "ssg":
            self._stream = _BatchStream(self, [synthetic.make_samples(seed + 1000 * rank + 17 * i, B, N, self.cat.name, dist_points)
                                               for i in range(int(stream_batches))])
            self.overlap = True      # the next batch's sampling rides on the same side stream as its collation
        # [B,3,N] as the loop feeds it (:207-208): a permuted VIEW of the collated [B,N,3] tensor, so the encoder's
        # permute back to points-major is free
        self.point_cloud = self.batch["point_cloud"].permute(0, 2, 1)
        self.prefetch = bool(prefetch_sampling) and self.device.type == "cuda"

    def forward_loss(self):
        return self._heads_loss(self._encode())

    class _Ticks:
        """Collect the num_batches_tracked counters of every train-mode BatchNorm the enclosed forward passes touch and advance
        them with one launch on exit (instead of one per set-abstraction level and one for the heads)."""

        def __init__(self, ts=None):
            self.ts = ts

        def __enter__(self):
            from . import sa_mlp
            self.prev, sa_mlp.DEFERRED_TICKS = sa_mlp.DEFERRED_TICKS, []

        def __exit__(self, *exc):
            from . import sa_mlp
            if self.ts is not None and exc[0] is None:
                self.ts._arm_and_tick()       # (the arena's launch advances the counters)
            else:
                sa_mlp.flush_ticks()
            sa_mlp.DEFERRED_TICKS = self.prev

    def _encode(self):
        # sa1's start is consumed only when its sampling was not prefetched; sa2's always
        sa1 = self.model.sa1
        _, r1, k1 = self._level_spec(sa1)
        multi = hasattr(sa1, "radius_list")
        ready = (self.prefetch or self.overlap) and pu.has_prefetched(self.batch["point_cloud"], sa1.npoint, r1 if multi else r1[0], k1 if multi else k1[0])
        starts = self.batch["fps_start"][1:] if ready else self.batch["fps_start"]
        if ready and self.overlap:
            starts = self.batch["fps_start"][len(self._plan_levels()):]   # every sampling level comes from the plan
        with pu.fps_start_override(starts):
            return self.model.encode(self.point_cloud)

    def _heads_loss(self, feat):
        out, sm_out, mask_conf, seg_conf = self.model.heads(feat)
        if self._drop_rng is not None:      # next step, next dropout masks: the step counter rides in the BatchNorm counters' launch
            from . import sa_mlp
            if sa_mlp.DEFERRED_TICKS is not None:
                sa_mlp.DEFERRED_TICKS.append(self._drop_step)
            else:
                self._drop_step.add_(1)
        return self.loss_handler.compute(return_list=False, y_pred=out, y=self.batch["traj"], pred_stroke_masks=sm_out,
                                         mask_scores=mask_conf, seg_logits=seg_conf, stroke_ids=self.batch["stroke_ids"],
                                         traj_as_pc=self.batch["traj_as_pc"])

    def _arm_and_tick(self):
        """Between the loss and its backward: the small zero-initialised outputs of the heads' / loss's / encoder's backward come out of
        one buffer cleared by ONE launch (ops.ZeroArena), and the same launch advances the step's counters -- every BatchNorm's
        num_batches_tracked, the dropout step, the dense optimizer's update count (three elementwise launches before)."""
        from . import sa_mlp
        ticks = list(sa_mlp.DEFERRED_TICKS or [])
        dense = getattr(self.opt, "step_dev", None)
        self._dense_ticked = False
        eager_opt = self.dp_graph and not self._dp_recorded        # (the dense optimizer of r5's data-parallel structure steps outside the graphs)
        if self._zero_arena is not None and self._zero_arena.arm(ticks, [dense] if (dense is not None and not eager_opt) else None):
            if sa_mlp.DEFERRED_TICKS:
                del sa_mlp.DEFERRED_TICKS[:]
            self._dense_ticked = dense is not None and not eager_opt
        else:
            sa_mlp.flush_ticks()

    def _opt_step(self):
        if isinstance(self.opt, DenseAdam):
            self.opt.step(ticked=self._dense_ticked)
        else:
            self.opt.step()
        self._dense_ticked = False

    def _disarm(self):
        if self._zero_arena is not None:
            self._zero_arena.disarm()

    def check(self, loss=None):
        """What the asynchronous step cannot do without a host sync: raise if the last stroke-mask matching failed (the reference
        asserts / scipy raises there: loss_handler.py:852-875) or the loss is not finite.  step() calls it every CHECK_EVERY
        steps; loops that log call it at their logging points."""
        self.loss_handler.check()
        loss = self._graph_loss if loss is None else loss
        if loss is not None and not bool(torch.isfinite(loss.detach()).all()):
            raise FloatingPointError(f"training loss is not finite ({float(loss.detach())}) after {self._steps_done} steps")

    CHECK_EVERY = int(os.environ.get("MASKPLANNER_CHECK_EVERY", "500"))   # 0: never from step()
    DP_GUARD_EVERY = 500   # replayed data-parallel steps: the replica guard (checksum all-gather + one host read) also every so many steps; 0: first steps only
    _steps_done = 0
    _adam_delay_cycles = 0
    SAMPLING_BEHIND_A = None   # set in __init__: MASKPLANNER_SAMPLING_BEHIND_A, default: multi-scale encoders
    GRAPH_AFTER = 3   # eager steps before recording (allocator warm, lazy kernel attributes set, optimizer state created)

    def step(self):
        """One optimisation step; returns the (device) loss tensor without synchronising (every CHECK_EVERY-th call reads the
        matching status and the loss of the step before: see check())."""
        loss = self._step()
        self._steps_done += 1
        if self.CHECK_EVERY and self._steps_done % self.CHECK_EVERY == 0:
            self.check(loss)
        return loss

    def _step(self):
        # [r4] host order at the step boundary.  The NEXT batch's plan is launched kernel by kernel on the second stream (a multi-scale
        # encoder with its plan extras: ~25 launches); issued BEFORE the replay of graph A that host time is a bubble on the step's
        # stream at every boundary (config 5: 0.75 ms under the tracer), issued entirely BEHIND it the first FPS starts late and graph B
        # waits for the plan instead (0.45 ms under the tracer).  So: the first level's FPS, then graph A, then the rest of the plan.
        # The single-scale encoder's ~10 launches keep the old order (same step time either way: NOTEBOOK.md).
        late = self.overlap and self._graph is not None and self._plan_cur is not None and self.SAMPLING_BEHIND_A and self._stream is None
        if self.overlap:
            # the NEXT batch's plan, on the second stream underneath this step (late: only its first kernel -- the first level's FPS, the
            # long pole of the plan -- before graph A; the other launches behind it)
            self._launch_sampling(phase="head" if late else None)
        if self._graph is not None:
            self._graph.replay()
            if late:
                self._launch_sampling(phase="tail")
            if self._graph_b is not None:
                if self._adam_ev is not None:
                    torch.cuda.current_stream().wait_event(self._adam_ev)   # the head weights of the previous step are final
                self._replay_b()
            elif self.overlap:
                self._hand_over()            # (single-graph step: the hand-over is not part of the recording)
            loss = self._graph_loss
            periodic = self.dp_graph and self.DP_GUARD_EVERY and self._guard_left <= 0 and (self._steps_done + 1) % self.DP_GUARD_EVERY == 0
            if (self._guard_left > 0 or periodic) and self.dp_graph:
                self._guard_left = max(self._guard_left - 1, 0)
                loss = loss.clone()           # (a fallback drops the graphs and their static loss tensor)
                self._dp_guard()
            return loss
        if not self.use_graph:
            return self._eager_step()
        if self._eager_steps < self.GRAPH_AFTER:
            # torch's recipe for capturing a training step: the iterations before the capture run on a side stream, so that
            # the autograd accumulator nodes and the caching allocator's blocks belong to a non-default stream
            self._eager_steps += 1
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                loss = self._eager_step()
            torch.cuda.current_stream().wait_stream(self._side)
            return loss
        self._record()
        return self._graph_loss if self._graph is not None else self._eager_step()

    def _record(self):
        """Record one full step into a hipGraph (torch.cuda.graph: private memory pool, graph-safe Philox offsets for the
        dropout layers).  The batch tensors, parameters and optimizer state are the static inputs.  Capture only records,
        so the graph is replayed once right away: this call performs exactly one optimisation step."""
        if (self._split_adam_wanted and self.factor_opt is not None) or self.dp_graph:
            return self._record_split()
        try:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with recording(g, **_capture_kw()):
                loss = self._eager_step(hand_over=False)
            self._graph, self._graph_loss = g, loss
            g.replay()
            if self.overlap:
                self._hand_over()
        except Exception as exc:   # stay correct: eager from here on
            import warnings
            warnings.warn(f"hipGraph capture of the training step failed ({type(exc).__name__}: {exc}); running eagerly")
            self._graph, self.use_graph = None, False
            torch.cuda.synchronize()

    def _record_split(self):
        """Two graphs on one capture stream and one memory pool (the autograd graph built while recording the first is walked
        while recording the second): A = encoder forward, B = heads + loss + backward + dense Adam.  The factor Adam is not
        recorded; its (x, g) factor buffers are static tensors of the pool, remembered here."""
        try:
            torch.cuda.synchronize()
            cap = torch.cuda.Stream()
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            if self.factor_opt is not None:
                self._reset_factor_store()
            from . import sa_mlp
            ticks_prev, sa_mlp.DEFERRED_TICKS = sa_mlp.DEFERRED_TICKS, []      # every BatchNorm counter of the step: one launch, in B
            kw = _capture_kw()
            with recording(ga, stream=cap, **kw):
                self._supply_plan()
                self.reducer.zero_grad()
                feat = self._encode()
            # The head optimizer runs OUTSIDE the graphs on its own stream, concurrently with the next replay of graph A.  Its
            # inputs therefore must not live in the graphs' memory pool: `feat` (the x factor of fc1 / sm_fc1) is rewritten by
            # graph A, and any pool block B allocated may be one A freed while it was recorded.  Graph B ends by copying the
            # factors (~5 MB) into buffers allocated here, outside the pool; only graph B writes them and it waits for the
            # optimizer's event before it is replayed.
            persist = self._alloc_factor_buffers() if self.factor_opt is not None else {}
            # With the head optimizer outside the graphs, B itself is recorded in two parts: B1 = heads + loss + the heads' own
            # backward (down to the gradient of the global feature), B2 = the encoder's backward + dense Adam.  The head
            # optimizer then starts right behind B1 and streams its ~1 GB underneath the encoder backward's matrix-core-bound
            # kernels instead of next to the HBM-bound first level of the following forward.
            # Default: only under data parallelism, where the factor all-gather then overlaps the backward as well.  On one GPU the
            # optimizer's ~1 GB costs the chain the same wherever it runs and the extra graph boundary costs ~60 us ([r2] 2.83 vs 2.76 ms).
            # [r6] Data parallel on RCCL: the bucket all-reduces and the dense Adam are RECORDED into the backward graph like the N = 1 step's
            # (ProcessGroupNCCL's collectives record under thread-local capture, as the SyncBN opt-in's do), so the N > 1 launch path is the
            # N = 1 path plus collective nodes: two graphs, then the head optimizer -- behind its factor all-gather -- on its own stream under
            # the next encoder forward.  No third graph boundary (the backward was split so that the all-gather could hide under the encoder's
            # backward; under the next forward it is hidden as well) and no eager exchange.  MASKPLANNER_DP_COLLECTIVES_GRAPH=0, or a
            # backend that cannot record (gloo: host-side collectives): r5's structure (three graphs, eager exchange and dense Adam).
            self._dp_recorded = bool(self.dp_graph and torch.distributed.get_backend() == "nccl"
                                     and os.environ.get("MASKPLANNER_DP_COLLECTIVES_GRAPH", "1") != "0")
            split_bwd = bool(persist) and os.environ.get("MASKPLANNER_SPLIT_BACKWARD", "1" if (dp.exchanging() and not self._dp_recorded) else "0") != "0"
            gb2 = torch.cuda.CUDAGraph() if split_bwd else None
            with recording(gb, pool=ga.pool(), stream=cap, **kw):
                if split_bwd:
                    leaf = feat.detach().requires_grad_(True)
                    loss = self._heads_loss(leaf)
                    self._arm_and_tick()
                    loss.backward(self._unit)     # (a cached 1: no fill launch for the seed gradient)
                else:
                    loss = self._heads_loss(feat)
                    self._arm_and_tick()
                    loss.backward(self._unit)     # (a cached 1: no fill launch for the seed gradient)
                    self._disarm()
                sa_mlp.DEFERRED_TICKS = ticks_prev
                if self.factor_opt is not None:
                    flush_bias_grads(self.model.factor_store)
                if (not self.dp_graph or self._dp_recorded) and not split_bwd:
                    self.reducer.finish()
                    self._opt_step()
                loss = loss.detach()
                if persist:
                    srcs, dsts, seen = [], [], {}
                    for k, (px, pg) in persist.items():
                        x, g = self.model.factor_store[k]
                        for s_, d_ in ((x, px), (g, pg)):
                            if d_.data_ptr() not in seen:
                                seen[d_.data_ptr()] = s_.data_ptr()
                                srcs.append(s_)
                                dsts.append(d_)
                            elif seen[d_.data_ptr()] != s_.data_ptr():
                                raise RuntimeError(f"factor {k}: expected to share its input activation")
                    torch._foreach_copy_(dsts, srcs)
                if self.overlap and not split_bwd:
                    self._hand_over_copies()
            if split_bwd:
                with recording(gb2, pool=ga.pool(), stream=cap, **kw):
                    feat.backward(leaf.grad)
                    self._disarm()
                    if not self.dp_graph or self._dp_recorded:
                        self.reducer.finish()
                        self._opt_step()
                    if self.overlap:
                        self._hand_over_copies()
            if self.dp_graph:
                self._static_grads = [(p, p.grad) for p in self.reducer.params if p.grad is not None]
            self._factor_args = persist
            if self.factor_opt is not None:
                self._reset_factor_store()
            self._adam_stream = torch.cuda.Stream()
            captured = (ga, gb, gb2, loss)
        except Exception as exc:   # stay correct: eager from here on
            import warnings
            from . import sa_mlp
            sa_mlp.DEFERRED_TICKS = None
            warnings.warn(f"hipGraph capture of the training step failed ({type(exc).__name__}: {exc}); running eagerly")
            captured = None
        # data parallelism: the launch mode is a COLLECTIVE decision -- a rank whose capture failed would exchange gradients from hooks
        # while the others replay graphs with the deferred exchange, and the mismatched collectives would hang the job
        ok = captured is not None
        if self.dp_graph:
            import torch.distributed as dist
            flag = torch.tensor([1.0 if ok else 0.0], device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item() > 0.5)
        if ok:
            self._graph, self._graph_b, self._graph_b2, self._graph_loss = captured
            self._graph.replay()
            self._replay_b()
        else:
            self._disarm()
            self._graph, self._graph_b, self._graph_b2, self.use_graph = None, None, None, False
            self._factor_args = None
            if self.dp_graph:
                self.dp_graph, self._guard_left, self.reducer.deferred, self._dp_recorded = False, 0, False, False
            if self.factor_opt is not None:
                self._reset_factor_store()      # queued bias-gradient entries point into the dropped graph pool
            torch.cuda.synchronize()

    @torch.no_grad()
    def _dp_guard(self):
        """Replica consistency of the graph-replayed data-parallel step: identical weights on every rank, all finite.  Collective (every
        rank calls it at the same step and sees the same gathered table, so every rank takes the same decision)."""
        import torch.distributed as dist
        if self._adam_ev is not None:
            torch.cuda.current_stream().wait_event(self._adam_ev)       # the head weights of this step are final
        ps = list(self.model.parameters())
        norms = torch.stack(torch._foreach_norm(ps)).double()
        wts = torch.arange(1, norms.numel() + 1, dtype=torch.float64, device=norms.device)
        mine = torch.stack([norms.sum(), (norms * wts).sum(), self._graph_loss.detach().double().reshape(())])
        world = dist.get_world_size()
        rows = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)                  # (the list form: every backend has it)
        table = torch.stack(rows).cpu()
        if not bool(torch.isfinite(table).any(dim=1).any()) or not bool(torch.isfinite(table[:, 2]).any()):
            # EVERY replica is non-finite: that is a diverged / poisoned run (e.g. a failed stroke-mask matching turns the loss into NaN
            # by design), not a replica mismatch -- raise with the reason instead of quietly re-broadcasting rank 0's state
            self.check(self._graph_loss)
            raise FloatingPointError("data-parallel step: parameters or loss are non-finite on every rank")
        ok = bool(torch.isfinite(table).all()) and bool((table[:, :2] == table[0, :2]).all())
        if os.environ.get("MASKPLANNER_DP_GUARD_TRIP") == str(self._guard_left):      # test hook: behave as if the replicas differed
            ok = False
        if ok:
            return
        import warnings
        warnings.warn("data-parallel graph replay: replicas differ or are non-finite after a replayed step "
                      f"(checksums {table[:, 0].tolist()}); taking rank 0's state and continuing with eager launches")
        self._dp_fallback()

    @torch.no_grad()
    def _dp_fallback(self):
        import torch.distributed as dist
        torch.cuda.synchronize()
        self._graph, self._graph_b, self._graph_b2, self._graph_loss = None, None, None, None
        self.use_graph, self.dp_graph, self._guard_left, self.dp_fell_back, self._dp_recorded = False, False, 0, True, False
        self.reducer.deferred = False
        self._static_grads, self._factor_args, self._adam_ev = None, None, None
        state = list(self.model.parameters()) + [b for b in self.model.buffers() if b.is_floating_point()]
        for opt in (self.opt, self.factor_opt):
            if opt is None:
                continue
            for st in opt.state.values():
                state += [v for v in st.values() if torch.is_tensor(v)]
            if getattr(opt, "step_dev", None) is not None:
                state.append(opt.step_dev)
        for t in state:
            dist.broadcast(t, src=0)
        for p in self.model.parameters():
            p.grad = None
        self.reducer.zero_grad()
        if self.factor_opt is not None:
            self._reset_factor_store()
        torch.cuda.synchronize()

    def _reset_factor_store(self):
        self.model.factor_store.clear()
        # head bias gradients: queued per Linear, reduced in one launch after backward().  Not when gradient hooks exchange
        # buckets DURING backward (eager data parallelism): a bucket would leave before its queued bias gradients exist.
        if not self.reducer.active or self.reducer.deferred:
            self.model.factor_store[BIAS_QUEUE] = []

    def _alloc_factor_buffers(self):
        """{key: (x [B,I], g [B,O])} persistent factor buffers for the deferred head optimizer, allocated by the ordinary
        caching allocator (call this outside any capture).  Weights that are fed by the same activation (fc3 / fc_normals,
        fc1 / sm_fc1) share one x buffer, as their factors in model.factor_store do (the all-gather de-duplicates by pointer)."""
        B = self.batch["point_cloud"].shape[0]
        shared_x = {"fc_normals.weight": "fc3.weight", "sm_fc1.weight": "fc1.weight"}
        out = {}
        for k, w in self.factor_opt.weights.items():
            O, I = w.shape
            src = shared_x.get(k)
            px = out[src][0] if src in out else torch.empty((B, I), dtype=torch.float32, device=self.device)
            out[k] = (px, torch.empty((B, O), dtype=torch.float32, device=self.device))
        return out

    def _replay_b(self):
        """Graph B (or B1, head optimizer, B2) and what follows it eagerly."""
        if self.overlap:
            self._wait_plan()                   # B's last node is the hand-over copy
        self._graph_b.replay()
        if self._graph_b2 is not None:
            self._launch_factor_adam()          # behind B1: underneath the encoder backward
            self._graph_b2.replay()
        if self.dp_graph and not self._dp_recorded:      # the exchange and the dense optimizer of a data-parallel step, eagerly on the step's stream
            self.reducer.rearm(self._static_grads)
            self.reducer.finish()
            self.opt.step()
        if self.factor_opt is not None and self._graph_b2 is None:
            self._launch_factor_adam()

    def _launch_factor_adam(self):
        side = self._adam_stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if self._adam_delay_cycles:      # test hook: hold the optimizer back so that it overlaps the next replay of graph A
                torch.cuda._sleep(int(self._adam_delay_cycles))
            self.model.factor_store.update(self._factor_args)
            self.factor_opt.step()
            self._adam_ev = torch.cuda.Event()
            self._adam_ev.record(side)

    def eager_step(self):
        """One step launched kernel by kernel even when a recorded graph exists (bench.py's per-kernel timing hooks live in
        the launch path).  Shares parameters and optimizer state with the graph, so the two can be interleaved.  Under data parallelism
        EVERY rank must take the same kind of step at the same time: a replayed step issues its collectives in another order (factor
        all-gather before the bucket all-reduces) than an eager one."""
        if self.overlap:
            self._launch_sampling()
        if self._adam_ev is not None:
            torch.cuda.current_stream().wait_event(self._adam_ev)
        return self._eager_step()

    def _plan_levels(self):
        """The sampling levels of the encoder (set abstractions that are not group_all), in order."""
        return [m for m in (self.model.sa1, self.model.sa2, self.model.sa3) if not getattr(m, "group_all", False)]

    @staticmethod
    def _level_spec(m):
        """(npoint, radii, group sizes) of a sampling level: one ball query for a single-scale level, one per radius for a multi-scale one."""
        if hasattr(m, "radius_list"):
            return m.npoint, list(m.radius_list), list(m.nsample_list)
        return m.npoint, [m.radius], [m.nsample]

    # [r4] What the loss derives from the TARGETS alone travels with the sampling plan: the padded lengths of the ground-truth segments
    # and points (pytorch3d_chamfer.py:138-149) and the screening planes of the nearest-neighbour search against the ground-truth
    # segments are computed for the NEXT batch on the second stream, next to its FPS, and handed over with the plan -- three launches
    # off the step's chain, still paid once per step (ops.register_static_target: the loss finds them by the target's address).
    TARGETS = (("traj", True), ("traj_as_pc", False))

    def _plan_targets(self):
        from . import ops
        out = []
        if self.device.type != "cuda":
            return out
        for k, planes in self.TARGETS:
            t = self.batch.get(k)
            if torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.ndim == 3:
                B, P2, D = t.shape
                # (every region of the plan buffer starts on a 16-byte boundary -- the screening workspace is read with 16-byte
                # loads, launch_knn1_screen refuses anything else --, so the B lengths in front of it occupy an even number of units)
                out.append((k, B, _even(B), _even((ops.target_aux_bytes(B, P2, D) + 7) // 8) if planes else 0))
        return out

    def _plan_extras(self):
        """Per-batch preprocessing of the encoder that depends on the cloud and its sampling alone, carried by the plan as well [r4]:
        ("gxyz", level, scale, int64 units): the first level's grouped, centred coordinate rows [B,S,K,4] (a level without input features;
        one entry per radius of a multi-scale level), rounded to bf16 values when the level's first layer would round them itself;
        ("rows", level, scale, units): the sorted row lists of a level whose first layer is factorised (sa_mlp.csr_rows)."""
        from . import sa_mlp
        out = []
        if self.device.type != "cuda":
            return out
        B = self.batch["point_cloud"].shape[0]
        levels = self._plan_levels()
        for li, m in enumerate(levels):
            S, _, Ks = self._level_spec(m)
            multi = hasattr(m, "radius_list")
            factored = sa_mlp.FACTORED_FIRST in (("1", "msg", True) if multi else ("1", True))
            for si, K in enumerate(Ks):
                if li == 0 and not getattr(self.model, "normal_channel", False):
                    out.append(("gxyz", li, si, B * S * K * 4 // 2))
                if li > 0 and factored and not self.sync_bn:
                    out.append(("rows", li, si, 2 * B * S * K // 2))
        return out

    def _plan_size(self):
        B, n = self.batch["point_cloud"].shape[0], 0
        for m in self._plan_levels():
            S, _, Ks = self._level_spec(m)
            n += B * S + (B * S * 3 + 1) // 2 + sum(B * S * K for K in Ks)
        n = _even(n)
        self._plan_aux_at = n
        for _, _, b, w in self._plan_targets():
            n += b + w
        self._plan_extra_at = n
        for _, _, _, units in self._plan_extras():
            n += _even(units)
        return n

    def _extra_views(self, buf):
        B, o, out = self.batch["point_cloud"].shape[0], self._plan_extra_at, []
        levels = self._plan_levels()
        for kind, li, si, units in self._plan_extras():
            S, _, Ks = self._level_spec(levels[li])
            K = Ks[si]
            v = buf[o:o + units]
            out.append((kind, li, si, v.view(torch.float32).view(B, S, K, 4) if kind == "gxyz" else v.view(torch.int32).view(2, B, S * K)))
            o += _even(units)
        return out

    def _rounds_gxyz(self, li, si):
        """The consumer of this level's grouped coordinates rounds them to bf16 values (sa_mlp.rounds_first_input): done here instead."""
        from . import sa_mlp
        m = self._plan_levels()[li]
        convs = m.conv_blocks[si] if hasattr(m, "radius_list") else m.mlp_convs
        return sa_mlp.rounds_first_input(convs, self._level_spec(m)[2][si], getattr(m, "mlp_dtype", "f32"))

    def _extras(self, buf, xyz=None):
        """Fill the extras of `buf` from ITS OWN sampling plan (same buffer: the views are consistent)."""
        from . import ops, sa_mlp
        xyz = self.batch["point_cloud"] if xyz is None else xyz
        plans = self._plan_views(buf)
        clouds = [xyz] + [p[1] for p in plans[:-1]]
        for kind, li, si, v in self._extra_views(buf):
            _, new_xyz, idxs = plans[li]
            if kind == "gxyz":
                ops.group_xyz_into(clouds[li], new_xyz, idxs[si], v)
                if self._rounds_gxyz(li, si):
                    v.copy_(v.to(torch.bfloat16))
            else:
                sa_mlp.csr_rows(idxs[si], clouds[li].shape[1], out=v)

    def _register_extras(self):
        from . import sa_mlp
        plans = self._plan_views(self._plan_cur)
        self._unregister_extras()
        for kind, li, si, v in self._extra_views(self._plan_cur):
            key = plans[li][2][si].data_ptr()
            if kind == "gxyz":
                pu._grouped_xyz[key] = v
                self._registered.append((pu._grouped_xyz, key))
                if self._rounds_gxyz(li, si):
                    sa_mlp.ROUNDED_INPUTS[v.data_ptr()] = True
                    self._registered.append((sa_mlp.ROUNDED_INPUTS, v.data_ptr()))
            else:
                sa_mlp.CSR_ROWS[key] = v
                self._registered.append((sa_mlp.CSR_ROWS, key))

    _registered = ()

    def _unregister_extras(self):
        """The registries are keyed by addresses inside this object's plan buffer (and hold views of it): entries go when the object goes,
        before the allocator can hand those addresses to somebody else."""
        for reg, key in self._registered:
            reg.pop(key, None)
        self._registered = []

    def __del__(self):
        try:
            self._unregister_extras()
        except Exception:
            pass

    def _target_views(self, buf):
        """Per target (key, lengths i64 [B], workspace u8 or None) as views of the plan buffer's tail."""
        o, out = self._plan_aux_at, []
        for k, B, b, w in self._plan_targets():
            out.append((k, buf[o:o + B], buf[o + b:o + b + w].view(torch.uint8) if w else None))
            o += b + w
        return out

    def _target_aux(self, buf, sources=None):
        from . import ops
        for k, lengths, ws in self._target_views(buf):
            ops.compute_target_aux(self.batch[k] if sources is None else sources[k], lengths, ws)

    def _register_targets(self):
        from . import ops
        for k, lengths, ws in self._target_views(self._plan_cur):
            ops.register_static_target(self.batch[k], planes=ws is not None, storage=(lengths, ws))

    def _plan_views(self, buf):
        """Per level (fps_idx i64 [B,S], new_xyz f32 [B,S,3], [idx i64 [B,S,K] per radius]) as views of one flat int64 buffer."""
        B, o, out = self.batch["point_cloud"].shape[0], 0, []
        for m in self._plan_levels():
            S, _, Ks = self._level_spec(m)
            n0, n1 = B * S, (B * S * 3 + 1) // 2
            fps_idx = buf[o:o + n0].view(B, S)
            new_xyz = buf[o + n0:o + n0 + n1].view(torch.float32)[:B * S * 3].view(B, S, 3)
            o += n0 + n1
            idxs = []
            for K in Ks:
                idxs.append(buf[o:o + B * S * K].view(B, S, K))
                o += B * S * K
            out.append((fps_idx, new_xyz, idxs))
        return out

    def _sample_levels(self, buf, xyz=None, starts=None, phase=None):
        """FPS + ball queries of every level: each level samples the previous level's centroids, nothing else -- the whole
        plan depends on the input cloud only.  phase "head": the first level's FPS alone; "tail": everything but that."""
        from . import ops
        xyz = self.batch["point_cloud"] if xyz is None else xyz
        starts = self.batch["fps_start"] if starts is None else starts
        for li, (m, start, (fps_idx, new_xyz, idxs)) in enumerate(zip(self._plan_levels(), starts, self._plan_views(buf))):
            _, radii, Ks = self._level_spec(m)
            if not (phase == "tail" and li == 0):
                start = torch.as_tensor(start, dtype=torch.long).to(xyz.device)
                ops.fps(xyz, m.npoint, start, out=(fps_idx, new_xyz))
            if phase == "head":
                return
            if len(radii) > 1:      # [r5] a multi-scale level: every radius in one scan of the cloud
                ops.ball_query_multi(radii, Ks, xyz, new_xyz, out=idxs)
            else:
                ops.ball_query(radii[0], Ks[0], xyz, new_xyz, out=idxs[0])
            xyz = new_xyz

    # Pipelined sampling, the protocol (the same for eager and replayed steps):
    #   start of a step : the NEXT batch's collation + sampling plan (FPS + ball query of every level) is launched on the second
    #                     stream into `_plan_next`, ordered after everything the step's stream holds (= the previous hand-over);
    #   end of a step   : the step's stream waits for that plan and copies it into `_plan_cur` (and, with streamed inputs, the staged
    #                     batch into the step's static tensors): the hand-over.  In the two-graph step the copy is the LAST NODE OF
    #                     GRAPH B and the wait sits in front of B's replay -- nothing is launched eagerly on the step's stream
    #                     between B and the next A.  [r4] measured: an eager kernel between two replays costs ~20 us before it and
    #                     ~80 us before the first kernel of the next graph (a 104 us bubble per step), a graph following a graph 13-19 us.
    # The sampling kernels themselves are never recorded: a second branch inside the hipGraph made the replay insert ~7 us
    # synchronisation gaps all along the main chain (26 per step).
    def _plan_init(self):
        """Both plan buffers; the FIRST batch's plan is computed in line on the step's stream."""
        self._plan_next = torch.zeros(self._plan_size(), dtype=torch.int64, device=self.device)
        self._plan_cur = torch.zeros_like(self._plan_next)
        self._plan_stream = torch.cuda.Stream()
        if self._stream is not None:
            xyz, starts = self._stream.collate_next()
            self._stream.collated()
            self._sample_levels(self._plan_cur, xyz, starts)
            self._target_aux(self._plan_cur, self._stream.stage)
            self._extras(self._plan_cur, xyz)
            self._stream.publish()
        else:
            self._sample_levels(self._plan_cur)
            self._target_aux(self._plan_cur)
            self._extras(self._plan_cur)
        self._register_targets()
        self._register_extras()

    def _launch_sampling(self, phase=None):
        """The next batch's collation + sampling plan on the second stream, ordered after everything the step's stream holds so
        far (at least the previous hand-over: only then may the next plan / the staging tensors be overwritten).
        phase "head": the ordering point and the first level's FPS only; "tail": everything behind it (resident batches)."""
        if self._plan_cur is None:
            self._plan_init()
        side = self._plan_stream
        if self._stream is not None and phase != "tail":
            with torch.cuda.stream(side):
                self._stream.prefetch()                          # host items of batch k + 2 -> pinned -> raw device set, BEFORE the wait below
        if phase != "tail":
            side.wait_stream(torch.cuda.current_stream())
        if self._stream is not None:
            # streamed inputs: ONE chain under the step (collation of the cloud -> FPS -> ball queries -> extras -> collation of the targets ->
            # their lengths / planes), device to device: the host-to-device transfer ran a step earlier on the copy stream (_BatchStream).
            # [r5] two chains on two streams (the targets beside the sampling) were measured: 2.38 vs 2.17 ms.
            with torch.cuda.stream(side):
                xyz, starts = self._stream.collate_cloud()       # batch k + 1: raw device set -> staging tensors
                self._sample_levels(self._plan_next, xyz, starts)
                self._extras(self._plan_next, xyz)
                self._stream.collate_targets()
                self._target_aux(self._plan_next, self._stream.stage)
                self._plan_ev = torch.cuda.Event()
                self._plan_ev.record(side)
            self._stream.collated(self._plan_ev)
            return
        with torch.cuda.stream(side):         # (resident inputs; the streamed case returned above)
            self._sample_levels(self._plan_next, phase=phase)
            if phase == "head":
                return
            self._target_aux(self._plan_next)
            self._extras(self._plan_next)
            self._plan_ev = torch.cuda.Event()
            self._plan_ev.record(side)

    def _wait_plan(self):
        if self._plan_ev is not None:
            torch.cuda.current_stream().wait_event(self._plan_ev)    # the next batch's sampling (and collation) is complete

    def _hand_over_copies(self):
        """The next plan becomes the step's (one elementwise kernel into the static buffer); with streamed inputs the collated next
        batch becomes the step's batch.  Static sources and destinations: recordable."""
        torch.add(self._plan_next, 0, out=self._plan_cur)
        if self._stream is not None:
            self._stream.publish()

    def _hand_over(self):
        self._wait_plan()
        self._hand_over_copies()

    def _supply_plan(self):
        if self.overlap:
            pu.clear_prefetched()    # nothing of an earlier (possibly aborted) step may survive into this one
            if self._plan_cur is None:
                self._plan_init()
            xyz = self.batch["point_cloud"]
            for m, (fps_idx, new_xyz, idxs) in zip(self._plan_levels(), self._plan_views(self._plan_cur)):
                if hasattr(m, "radius_list"):
                    pu.supply_sampling(xyz, m.npoint, m.radius_list, m.nsample_list, (fps_idx, new_xyz, tuple(idxs)))
                else:
                    pu.supply_sampling(xyz, m.npoint, m.radius, m.nsample, (fps_idx, new_xyz, idxs[0]))
                xyz = new_xyz    # the next level's cloud IS this level's centroid tensor (same storage: the lookup key)

    def _eager_step(self, hand_over=True):
        try:
            loss = self._eager_step_body()
            if hand_over and self.overlap and self._plan_ev is not None:
                self._hand_over()
            return loss
        except BaseException:
            pu.clear_prefetched()    # a plan queued for this step must not outlive it
            raise

    def _eager_step_body(self):
        self._supply_plan()
        self.reducer.zero_grad()
        with self._Ticks(self):
            loss = self.forward_loss()
        try:
            loss.backward(self._unit)     # (a cached 1: no fill launch for the seed gradient)
        finally:
            self._disarm()
        if self.factor_opt is not None:
            flush_bias_grads(self.model.factor_store)
        self.reducer.finish()
        if self.prefetch:
            # the next batch (here: the same synthetic one) is already resident: run ITS first-level FPS + ball query on
            # the side stream while the optimizer's bandwidth-bound streaming kernels occupy the main stream
            sa1 = self.model.sa1
            pu.prefetch_sampling(self.batch["point_cloud"], sa1.npoint, sa1.radius, sa1.nsample, self.batch["fps_start"][0])
        self._opt_step()
        if self.factor_opt is not None:
            self.factor_opt.step()
        return loss.detach()   # callers never keep the autograd graph (and its accumulator nodes) alive across steps


class _BatchStream:
    """Host dataset items -> the step's device batch (TrainStep(stream_batches=K)), as a two-stage pipeline [r5]:
      upload(k+2)    host items -> a pinned set (numpy) -> a RAW device set (flat ragged rows, offsets, FPS start indices) by hipMemcpyAsync on
                     the plan stream, IN FRONT of that stream's wait for step k: the PCIe transfer (7.5 MB per cuboids batch, ~0.15 ms) runs in
                     the stream's idle tail of step k - 1, on no chain of any step;
      collate(k+1)   on the plan stream, under step k: raw set -> the staging tensors of the step's fixed shapes (csrc/collate.hip's pad
                     kernel, device to device), then FPS / ball queries / target lengths + planes (harness._launch_sampling);
      publish()      staging -> the step's static tensors, the last node of graph B.
    With the transfer INSIDE the plan chain (r4, and the zero-copy / two-chain variants measured this round: tools/stream_gap.py) the plan of
    batch k+1 was ready ~0.13 ms after graph A of step k had finished and graph B waited for it: 2.17 vs 2.03 ms per step."""
    KEYS = (("traj", -100.0), ("traj_as_pc", -100.0), ("stroke_ids", -1.0))
    NSETS = 2

    def __init__(self, ts, batches):
        self.ts, self.batches = ts, batches
        self.n_up, self.n_co = 0, 0
        dev = ts.device
        B, N = ts.batch["point_cloud"].shape[:2]
        self.width = {k: max(max(it[k].shape[0] for it in b) for b in batches) for k, _ in self.KEYS}
        self.dim = {k: (batches[0][0][k].shape[1] if batches[0][0][k].ndim == 2 else 1) for k, _ in self.KEYS}
        # the step's static tensors get the fixed widths (before anything is recorded)
        for k, fill in self.KEYS:
            shape = (B, self.width[k]) + ((self.dim[k],) if batches[0][0][k].ndim == 2 else ())
            ts.batch[k] = torch.full(shape, fill, dtype=torch.float32, device=dev)
        self.stage = {k: torch.empty_like(ts.batch[k]) for k, _ in self.KEYS}
        self.stage["point_cloud"] = torch.empty_like(ts.batch["point_cloud"])
        self.stage_starts = [torch.zeros(B, dtype=torch.long, device=dev) for _ in ts.batch["fps_start"]]
        cap = {k: B * self.width[k] * self.dim[k] for k, _ in self.KEYS}
        self.sets = []
        for _ in range(self.NSETS):
            pin = {k: torch.empty(cap[k], dtype=torch.float32).pin_memory() for k, _ in self.KEYS}
            pin["point_cloud"] = torch.empty(B * N * 3, dtype=torch.float32).pin_memory()
            raw = {k: torch.empty(cap[k], dtype=torch.float32, device=dev) for k, _ in self.KEYS}
            raw["point_cloud"] = torch.empty_like(ts.batch["point_cloud"])
            self.sets.append(dict(pin=pin, raw=raw, tot={},
                                  pin_off={k: torch.empty(B + 1, dtype=torch.int64).pin_memory() for k, _ in self.KEYS},
                                  raw_off={k: torch.empty(B + 1, dtype=torch.int64, device=dev) for k, _ in self.KEYS},
                                  pin_starts=[torch.empty(B, dtype=torch.long).pin_memory() for _ in ts.batch["fps_start"]],
                                  raw_starts=[torch.zeros(B, dtype=torch.long, device=dev) for _ in ts.batch["fps_start"]],
                                  ev_h2d=None, ev_read=None))
        self.levels_n = [N] + [m.npoint for m in ts._plan_levels()][:-1]
        self._cur = None

    def upload(self):
        """The next host batch into a pinned set and on to its raw device set, on the CURRENT stream (host-side numpy + seven asynchronous
        copies).  The harness calls it on the plan stream BEFORE that stream's wait for the step: the copies then run right behind the
        previous plan chain, in the stream's idle tail of the previous step, and need no stream or event of their own.  (A fourth busy stream
        -- a copy stream of its own was the first version -- shares one of this runtime's four hardware queues with another stream of the
        step: 2.32 ms per step instead of 2.06; GPU_MAX_HW_QUEUES=16 gave 2.10.)"""
        import numpy as np
        items = self.batches[self.n_up % len(self.batches)]
        hs = self.sets[self.n_up % self.NSETS]
        self.n_up += 1
        if hs["ev_h2d"] is not None:
            hs["ev_h2d"].synchronize()      # (two uploads ago: long done) before this pinned set is rewritten
        pin, B = hs["pin"], len(items)
        np.stack([it["point_cloud"] for it in items], out=pin["point_cloud"].numpy().reshape(B, -1, 3))
        for k, _fill in self.KEYS:
            arrs = [np.asarray(it[k], dtype=np.float32).reshape(it[k].shape[0], -1) for it in items]
            lens = [a.shape[0] for a in arrs]
            tot = sum(lens)
            np.concatenate(arrs, axis=0, out=pin[k].numpy()[:tot * self.dim[k]].reshape(tot, self.dim[k]))
            off = hs["pin_off"][k].numpy()
            off[0] = 0
            np.cumsum(lens, out=off[1:])
            hs["tot"][k] = tot
        for p_, n in zip(hs["pin_starts"], self.levels_n):      # the reference's draw (pointnet2_utils.py:77)
            p_.copy_(torch.randint(0, n, (B,), dtype=torch.long))
        cur = torch.cuda.current_stream()
        if hs["ev_read"] is not None:
            cur.wait_event(hs["ev_read"])       # the raw set's previous content has been collated (same stream in steady state: no-op)
        hs["raw"]["point_cloud"].copy_(pin["point_cloud"].view_as(hs["raw"]["point_cloud"]), non_blocking=True)
        for k, _fill in self.KEYS:
            n = hs["tot"][k] * self.dim[k]
            hs["raw"][k][:n].copy_(pin[k][:n], non_blocking=True)
            hs["raw_off"][k].copy_(hs["pin_off"][k], non_blocking=True)
        for p_, d_ in zip(hs["pin_starts"], hs["raw_starts"]):
            d_.copy_(p_, non_blocking=True)
        hs["ev_h2d"] = torch.cuda.Event()
        hs["ev_h2d"].record(cur)

    def prefetch(self):
        """Keep the upload one batch ahead of the collation (call at the start of a step, before the plan chain is launched)."""
        while self.n_up < self.n_co + 2:
            self.upload()

    def collate_next(self):
        """Both halves on the current stream (the first batch, in line)."""
        out = self.collate_cloud()
        self.collate_targets()
        return out

    def collate_cloud(self):
        """The next uploaded batch's cloud rows and FPS start indices into the staging tensors, on the current stream (device to device)."""
        if self.n_up <= self.n_co:
            self.upload()
        hs = self._cur = self.sets[self.n_co % self.NSETS]
        self.n_co += 1
        torch.cuda.current_stream().wait_event(hs["ev_h2d"])
        torch._foreach_copy_([self.stage["point_cloud"]] + list(self.stage_starts), [hs["raw"]["point_cloud"]] + list(hs["raw_starts"]))
        return self.stage["point_cloud"], self.stage_starts

    def collate_targets(self):
        from . import _lib, ops
        lib = _lib.load()
        hs = self._cur
        B = self.stage["point_cloud"].shape[0]
        for k, fill in self.KEYS:
            ops._run("pad_ragged", self.stage[k], lib.mp_pad_ragged_f32, hs["raw"][k].data_ptr(), hs["raw_off"][k].data_ptr(), B,
                     self.width[k], self.dim[k], float(fill), self.stage[k].data_ptr())

    def collated(self, ev=None):
        """ev: the event behind which the current raw set has been read (None: recorded here, on the current stream)."""
        if ev is None:
            ev = torch.cuda.Event()
            ev.record()
        self._cur["ev_read"] = ev

    def publish(self):
        b = self.ts.batch
        torch._foreach_copy_([b["point_cloud"]] + [b[k] for k, _ in self.KEYS], [self.stage["point_cloud"]] + [self.stage[k] for k, _ in self.KEYS])


class DropInLoop:
    """The reference's own hot loop (train_maskplanner.py:182-227), statement for statement, on the drop-in modules -- what an
    UNCHANGED train_maskplanner.py executes per batch once `dropin.install()` has aliased the modules: the model the
    reference's factory builds (models/__init__.py:111-122), `torch.optim.Adam` over ALL parameters (:159), a fresh collated
    HOST batch every step (copied to the device inside the loop, :207-208; the GT tensors inside the loss, loss_handler.py:
    629, 838), FPS start indices drawn from the CPU generator per call (pointnet2_utils.py:77), `compute()` returning the
    numpy list of terms and `loss.item()` (:212-224) -- two host synchronisations per step.  No factor heads, no pipelined sampling, no fused
    optimizer: this is the drop-in figure, `TrainStep` is the path's ceiling.  ([r5] What the loop cannot see is how `model(...)` and
    `loss_handler.compute(...)` launch their work: after three calls per shape they replay graphs recorded from their own eager code,
    maskplanner_amd/graphed.py.)

    [r6] The host batches are what the reference's collate produces for the maskplanner alias (traj_with_equally_spaced_points:
    configs/maskplanner/traj_sampling_v2.yaml:9): every batch padded to ITS OWN maximum (utils/dataset/paintnet_ODv1.py:738-747), so `traj`,
    `stroke_ids` and `traj_as_pc` change width from batch to batch.  `n_batches` (default 32) of them, visited in a shuffled order per pass
    like the loop's `DataLoader(shuffle=True)`; `widths()` says how many distinct shapes the pool holds, `graph_stats()` how often the two
    calls replayed."""

    def __init__(self, category="cuboids", B=32, N=5120, device="cuda", seed=1235, lr=1e-3, dist_points="cuboid", n_batches=32,
                 rank=0, adam_kwargs=None):
        self.cat = synthetic.CATEGORIES[category] if isinstance(category, str) else category
        self.device = torch.device(device)
        torch.manual_seed(seed)
        self.model = maskplanner_model(self.cat).to(self.device)
        self.opt = torch.optim.Adam(self.model.parameters(), lr=lr, **(adam_kwargs or {}))    # :159 (adam_kwargs: what a one-word change
                                                                                          # of that line buys, e.g. fused=True -- tools/dropin_phases.py)
        self.cfg = maskplanner_loss_config()
        self.loss_handler = LossHandler(["asymm_v6_chamfer_with_stroke_masks"], self.cfg)  # :163
        self.host_batches = [synthetic.make_batch(seed + 1000 * rank + 7 * i, B, N, self.cat.name, dist_points) for i in range(n_batches)]
        self._i = 0
        self._order = list(range(n_batches))
        self._rng = np.random.default_rng(seed + 17)
        self.tot_loss, self.tot_loss_list = 0.0, 0.0

    def widths(self):
        """The distinct (n_segments, n_points) paddings of the pool's batches."""
        return sorted({(int(b["traj"].shape[1]), int(b["traj_as_pc"].shape[1])) for b in self.host_batches})

    def graph_stats(self):
        from . import graphed
        return {"model": graphed.stats(self.model), "loss": graphed.loss_stats(self.loss_handler)}

    def step(self):
        """One iteration of the loop body; returns the host float `loss.item()`."""
        n = len(self.host_batches)
        if self._i % n == 0 and n > 1:
            self._rng.shuffle(self._order)        # a new pass over the pool: the DataLoader's shuffle (train_maskplanner.py:135-141)
        data = self.host_batches[self._order[self._i % n]]
        self._i += 1
        model, device = self.model, self.device
        if self._i == 1:
            model.train()            # (:181: once per epoch, in front of the batch loop)
        model.zero_grad()
        point_cloud, traj = data["point_cloud"], data["traj"]
        traj_as_pc, stroke_ids = data["traj_as_pc"], data["stroke_ids"]
        B = point_cloud.shape[0]
        point_cloud = point_cloud.permute(0, 2, 1)
        point_cloud, traj = point_cloud.to(device, dtype=torch.float), traj.to(device, dtype=torch.float)
        traj_pred, pred_stroke_masks, mask_scores, seg_logits = model(point_cloud)
        loss, loss_list = self.loss_handler.compute(y_pred=traj_pred, y=traj, pred_stroke_masks=pred_stroke_masks,
                                                    mask_scores=mask_scores, seg_logits=seg_logits, stroke_ids=stroke_ids,
                                                    traj_as_pc=traj_as_pc)
        loss.backward()
        self.opt.step()
        value = loss.item()
        self.tot_loss += value * B
        self.tot_loss_list = self.tot_loss_list + np.asarray(loss_list) * B
        del point_cloud, traj, traj_pred, mask_scores, seg_logits, pred_stroke_masks
        model.zero_grad()
        return value
