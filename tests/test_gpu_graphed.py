"""The drop-in model's forward / backward replayed from recorded graphs inside an unchanged loop (maskplanner_amd/graphed.py) against the same
module code launched op by op: same outputs, same parameter gradients, same BatchNorm statistics from the same seeds; the cases that must
stay eager do; gradient accumulation and a second forward before a backward keep autograd's meaning."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from maskplanner_amd import _lib, graphed, pointnet2_cls_ssg as pc, pointnet2_utils as pu, synthetic
    _lib.load()
    return graphed, pc, pu, synthetic


def _model(pc, synthetic, seed=3, hidden=(256, 256)):
    torch.manual_seed(seed)
    return pc.maskplanner_model(synthetic.CATEGORIES["cuboids"], hidden_size=hidden).cuda().train()


def _clouds(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, N, 3, generator=g) * 2 - 1).permute(0, 2, 1).cuda()      # what the loop hands over: a permuted [B, N, 3]


def _step(model, x, gouts, seed):
    """One forward + backward with given output gradients; the CPU generator (FPS starts) and the device generator (dropout) seeded."""
    torch.manual_seed(seed)
    outs = model(x)
    keep = [o for o in outs if o is not None]
    torch.autograd.backward(keep, [g for g in gouts if g is not None])
    return [o.detach().clone() for o in keep]


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12))


def _worst_grad(model, want):
    """Largest gradient difference over the parameters, each relative to its own largest entry -- or, for the parameters whose true gradient is
    zero (a bias in front of a BatchNorm: rounding noise only), to a thousandth of the largest gradient entry of the model."""
    gmax = max(float(g.abs().max()) for g in want.values() if g is not None)
    worst = 0.0
    for n, p in model.named_parameters():
        assert (p.grad is None) == (want[n] is None), n
        if p.grad is not None:
            d = float((p.grad.double() - want[n].double()).abs().max())
            worst = max(worst, d / max(float(want[n].abs().max()), 1e-3 * gmax))
    return worst


def test_replayed_step_equals_the_eager_step(mods):
    graphed, pc, pu, synthetic = mods
    B, N = 4, 1024
    eager, rec = _model(pc, synthetic), _model(pc, synthetic)
    xs = [_clouds(B, N, 10 + i) for i in range(6)]
    with torch.no_grad():
        probe = [o for o in eager.eval()(xs[0]) if o is not None]
    eager.train()
    torch.manual_seed(0)
    gouts = [torch.randn_like(o) for o in probe]
    for i, x in enumerate(xs):
        for m, on in ((eager, False), (rec, True)):
            graphed.ENABLED = on
            m.zero_grad()
            try:
                outs = _step(m, x, gouts, 100 + i)
            finally:
                graphed.ENABLED = True
            if on:
                got = outs
            else:
                want = outs
        runner = next(iter(rec._graph_runners.values()))
        assert (runner.graph_r is not None) == (i >= graphed.WARM)         # recorded on the fourth call, replayed from then on
        for a, b in zip(got, want):
            assert _rel(a, b) < 2e-5, (i, _rel(a, b))
        worst = _worst_grad(rec, {n: q.grad for n, q in eager.named_parameters()})
        assert worst < 2e-3, (i, worst)                                     # (atomics in the weight-gradient kernels: run-to-run rounding)
        for (n, b), c in zip(rec.named_buffers(), eager.buffers()):
            if b.dtype.is_floating_point:
                assert _rel(b, c) < 1e-5, n
            else:
                assert torch.equal(b, c), n                                 # num_batches_tracked ticks inside the graph too


def test_accumulation_and_second_forward_keep_autograd_meaning(mods):
    graphed, pc, pu, synthetic = mods
    B, N = 4, 1024
    m = _model(pc, synthetic, seed=5)
    x = _clouds(B, N, 40)
    for i in range(graphed.WARM + 1):                                       # past the recording
        m.zero_grad()
        out = m(x)
        out[0].sum().backward()
    assert next(iter(m._graph_runners.values())).graph_r is not None
    # two backward passes without zero_grad: the gradients add (the first pass's values live in the static buffers the second one rewrites)
    m.zero_grad()
    torch.manual_seed(1); m(x)[0].sum().backward()
    g1 = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    torch.manual_seed(2); m(x)[0].sum().backward()
    both = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad()
    torch.manual_seed(2); m(x)[0].sum().backward()
    second = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    for n, p in m.named_parameters():
        if p.grad is not None:
            p.grad = both[n]
    worst = _worst_grad(m, {n: (g1[n] + second[n] if n in g1 else None) for n, _ in m.named_parameters()})
    assert worst < 2e-3, worst
    # a second forward before the first one's backward runs eagerly and both can be backpropagated
    m.zero_grad()
    a = m(x)[0]
    b = m(x)[0]
    assert type(a.grad_fn).__name__.startswith("_Replay") and not type(b.grad_fn).__name__.startswith("_Replay")
    (a.sum() + b.sum()).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    # a backward through a forward whose buffers were reused is an error, not stale numbers
    m.zero_grad()
    a = m(x)[0]
    a.sum().backward()
    with pytest.raises(RuntimeError):
        a.sum().backward()


def test_what_stays_eager_and_what_is_recorded_separately(mods):
    graphed, pc, pu, synthetic = mods
    m = _model(pc, synthetic, seed=6)
    x = _clouds(4, 1024, 50)
    # evaluation WITH autograd on (nobody backpropagates, but somebody could): eager
    m.eval()
    for _ in range(graphed.WARM + 2):
        o = m(x)[0]
        assert not type(o.grad_fn).__name__.startswith("_Replay")
        del o
    assert not m.__dict__.get("_graph_runners")
    # evaluation under no_grad: a forward-only graph of its own; the outputs equal the eager evaluation for the same FPS starts
    m.eval()
    with torch.no_grad():
        for i in range(graphed.WARM + 2):
            torch.manual_seed(9)
            got = m(x)
        graphed.ENABLED = False
        try:
            torch.manual_seed(9)
            want = m(x)
        finally:
            graphed.ENABLED = True
    r = [r for k, r in m._graph_runners.items() if not k[3]]
    assert len(r) == 1 and r[0].graph_f is not None and r[0].graph_r is None
    for a, b in zip(got, want):
        if a is not None:
            assert _rel(a, b) < 1e-5
    # another batch size: eager until it has been seen WARM times, then its own recording; a copy of the model starts without graphs
    m.train()
    x2 = _clouds(2, 1024, 51)
    for i in range(graphed.WARM + 1):
        m.zero_grad()
        o = m(x2)[0]
        assert type(o.grad_fn).__name__.startswith("_Replay") == (i >= graphed.WARM)
        o.sum().backward()
    assert not copy.deepcopy(m).__dict__.get("_graph_runners")
    graphed.reset(m)
    assert not m.__dict__.get("_graph_runners")


def test_the_loop_body_on_graphs_trains_like_the_eager_loop(mods):
    """harness.DropInLoop (the reference's loop body, statement for statement): 30 steps with and without the recorded graphs from one seed --
    both fall, and stay within the spread two eager runs show (atomics make the trajectories diverge after a few steps)."""
    graphed, pc, pu, synthetic = mods
    from maskplanner_amd.harness import DropInLoop
    runs = {}
    for on in (False, True):
        graphed.ENABLED = on
        try:
            torch.manual_seed(11)
            loop = DropInLoop("cuboids", B=8, N=1024, n_batches=4)       # (four batches in rotation: the loss of a pass is comparable to the last one's)
            runs[on] = [loop.step() for _ in range(30)]
            if on:
                assert any(r.graph_r is not None for r in loop.model._graph_runners.values())
        finally:
            graphed.ENABLED = True
    for on in (False, True):
        v = runs[on]
        assert np.isfinite(v).all() and np.mean(v[-5:]) < 0.9 * np.mean(v[:3]), v
    assert abs(runs[True][0] - runs[False][0]) < 1e-3 * abs(runs[False][0])          # the first steps are the same eager code
    assert abs(np.mean(runs[True][-5:]) - np.mean(runs[False][-5:])) < 0.25 * np.mean(runs[False][-5:])


def test_replayed_loss_equals_the_eager_loss(mods):
    """LossHandler.compute(...) of the loop: values, term list and the gradients of the model's outputs, recorded graphs against eager launches,
    on changing batches; a config change (the reference reschedules loss weights between epochs) gets a recording of its own."""
    graphed, pc, pu, synthetic = mods
    from maskplanner_amd.loss_handler import LossHandler, maskplanner_loss_config
    cat = synthetic.CATEGORIES["cuboids"]
    B, N = 4, 1024
    torch.manual_seed(21)
    model = pc.maskplanner_model(cat, hidden_size=(256, 256)).cuda().eval()
    handlers = {on: LossHandler(["asymm_v6_chamfer_with_stroke_masks"], maskplanner_loss_config()) for on in (False, True)}
    W = graphed.WARM
    data = synthetic.make_batch(300, B, N, cat.name, "cuboid")       # (one batch: the widths repeat; the ragged case is the next test)
    x = data["point_cloud"].permute(0, 2, 1).cuda().float()
    with torch.no_grad():
        base = model(x)
    for i in range(2 * W + 4):
        torch.manual_seed(500 + i)
        outs = [None if o is None else o + 0.05 * torch.randn_like(o) for o in base]
        if i == W + 2:
            for h in handlers.values():
                h.config["weight_asymm_v6_chamfer_with_stroke_masks"] = 0.5          # another epoch's weight
        res = {}
        for on in (False, True):
            graphed.ENABLED = on
            try:
                leaves = [o.detach().clone().requires_grad_(True) if o is not None else None for o in outs]
                loss, terms = handlers[on].compute(y_pred=leaves[0], y=data["traj"].cuda().float(), pred_stroke_masks=leaves[1], mask_scores=leaves[2],
                                                   seg_logits=leaves[3], stroke_ids=data["stroke_ids"], traj_as_pc=data["traj_as_pc"])
                (2.0 * loss).backward()
            finally:
                graphed.ENABLED = True
            res[on] = (loss.detach(), terms, [l.grad for l in leaves if l is not None])
            if on:
                # recorded on the fourth call; the rescheduled weight is recorded at its FIRST sighting (the eager warm-up is per argument
                # signature, not per config)
                assert type(loss.grad_fn).__name__.startswith("_LossReplay") == (i >= W), i
        assert _rel(res[True][0], res[False][0]) < 1e-5
        assert np.allclose(res[True][1], res[False][1], rtol=1e-5)
        for a, b in zip(res[True][2], res[False][2]):
            assert (a is None) == (b is None)
            if a is not None:
                assert _rel(a, b) < 1e-4
    assert len(handlers[True]._graph_runners) == 2          # the two weights
    st = graphed.loss_stats(handlers[True])
    assert st["recorded"] == 2 and st["eager"] == W and st["replayed"] == W + 2 and st["evicted"] == 0, st


def _narrow(synthetic, lo, hi):
    """cuboids with the ground-truth pose count of a sample drawn from [lo, hi]: the batch maxima (= the collate's padding widths) fall where
    the test wants them."""
    import dataclasses
    return dataclasses.replace(synthetic.CATEGORIES["cuboids"], points_lo=lo, points_hi=hi)


def _loss_inputs(model, data, seed):
    x = data["point_cloud"].permute(0, 2, 1).cuda().float()
    with torch.no_grad():
        base = model(x)
    torch.manual_seed(seed)
    return [None if o is None else o + 0.05 * torch.randn_like(o) for o in base]


def _compute(handler, outs, data):
    leaves = [o.detach().clone().requires_grad_(True) if o is not None else None for o in outs]
    y = data["traj"].cuda().float()          # (the loop's own statement, train_maskplanner.py:208; stroke_ids / traj_as_pc stay host tensors)
    loss, terms = handler.compute(y_pred=leaves[0], y=y, pred_stroke_masks=leaves[1], mask_scores=leaves[2], seg_logits=leaves[3],
                                  stroke_ids=data["stroke_ids"], traj_as_pc=data["traj_as_pc"])
    loss.backward()
    return loss, terms, [l.grad for l in leaves if l is not None]


def test_ragged_ground_truth_widths_share_recordings(mods, monkeypatch):
    """[r6] The reference's collate pads every batch to its own maximum (utils/dataset/paintnet_ODv1.py:738-747; traj_sampling_v2.yaml:9): 16
    batches with 16 distinct (n_segments, n_points) widths, shuffled, twice over.  At most two recordings (one per capacity bucket the widths
    fall into), everything after them replays, and every call's loss and gradients equal the eager call's."""
    graphed, pc, pu, synthetic = mods
    from maskplanner_amd import ops
    from maskplanner_amd.loss_handler import LossHandler, maskplanner_loss_config
    monkeypatch.setattr(ops, "DETERMINISTIC", True)      # the scatter backward of the reverse terms in its fixed-order form: its atomics' run-to-run
                                                         # noise (1.2e-6 of the largest gradient entry between two EAGER calls) is not what is compared
    B, N = 4, 1024
    torch.manual_seed(31)
    model = pc.maskplanner_model(synthetic.CATEGORIES["cuboids"], hidden_size=(256, 256)).cuda().eval()
    pool, widths = [], set()
    seed = 700
    while len(pool) < 16:
        # 13 batches whose pose maxima fall into (2816, 2944], three into (2944, 3072]; the segment maxima all into (896, 1024]
        cat = _narrow(synthetic, 2830, 2940) if len(pool) < 13 else _narrow(synthetic, 2950, 3060)
        d = synthetic.make_batch(seed, B, N, cat, "cuboid")
        seed += 1
        w = (d["traj"].shape[1], d["traj_as_pc"].shape[1])
        if w not in widths:
            widths.add(w)
            pool.append(d)
    assert len(widths) == 16 and len({w[0] for w in widths}) > 4 and len({w[1] for w in widths}) > 8
    order = list(np.random.default_rng(5).permutation(16)) + list(np.random.default_rng(6).permutation(16))
    handlers = {on: LossHandler(["asymm_v6_chamfer_with_stroke_masks"], maskplanner_loss_config()) for on in (False, True)}
    replayed = 0
    for step, j in enumerate(order):
        data = pool[j]
        outs = _loss_inputs(model, data, 900 + step)
        res = {}
        for on in (False, True):
            graphed.ENABLED = on
            try:
                res[on] = _compute(handlers[on], outs, data)
            finally:
                graphed.ENABLED = True
        replayed += type(res[True][0].grad_fn).__name__.startswith("_LossReplay")
        assert _rel(res[True][0], res[False][0]) < 1e-6, (step, _rel(res[True][0], res[False][0]))
        assert np.allclose(res[True][1], res[False][1], rtol=1e-6)
        for a, b in zip(res[True][2], res[False][2]):
            assert (a is None) == (b is None)
            if a is not None:
                assert _rel(a, b) < 1e-6, (step, _rel(a, b))
    st = graphed.loss_stats(handlers[True])
    assert st["recorded"] <= 2 and st["evicted"] == 0, st
    assert st["replayed"] + st["recorded"] == replayed and st["replayed"] >= len(order) - graphed.WARM - 2, st
    assert all(c["traj_as_pc"] in (2944, 3072) and c["y"] == 1024 and c["stroke_ids"] == 1024 for c in st["capacities"]), st


def test_loss_recordings_are_evicted_least_recently_used(mods):
    """More configs than MAX_LOSS_KEYS (the reference reschedules loss weights between epochs): the recording that was USED longest ago goes,
    not the one that was recorded first; a call whose shapes were seen once is not recorded."""
    graphed, pc, pu, synthetic = mods
    from maskplanner_amd.loss_handler import LossHandler, maskplanner_loss_config
    B, N = 4, 1024
    torch.manual_seed(32)
    model = pc.maskplanner_model(synthetic.CATEGORIES["cuboids"], hidden_size=(256, 256)).cuda().eval()
    data = synthetic.make_batch(310, B, N, "cuboids", "cuboid")
    outs = _loss_inputs(model, data, 1)
    lh = LossHandler(["asymm_v6_chamfer_with_stroke_masks"], maskplanner_loss_config())
    K = graphed.MAX_LOSS_KEYS

    def call(weight):
        lh.config["weight_asymm_v6_chamfer_with_stroke_masks"] = weight
        loss, _, _ = _compute(lh, outs, data)
        return loss

    def recorded_weights():
        return sorted(float(dict(k[0][2])["weight_asymm_v6_chamfer_with_stroke_masks"]) for k, r in lh._graph_runners.items() if r.graph_l is not None)
    for _ in range(graphed.WARM):
        call(1.0)
    for w in range(1, K + 1):                       # K recordings: weights 1 .. K
        assert type(call(float(w)).grad_fn).__name__.startswith("_LossReplay")
    assert recorded_weights() == [float(w) for w in range(1, K + 1)]
    call(1.0)                                       # weight 1 is the most recently used now
    call(float(K + 1))                              # one more than fits: the least recently used (2) goes
    assert recorded_weights() == sorted([1.0] + [float(w) for w in range(3, K + 2)])
    st = graphed.loss_stats(lh)
    assert st["evicted"] == 1 and st["recorded"] == K + 1, st
    ref = LossHandler(["asymm_v6_chamfer_with_stroke_masks"], maskplanner_loss_config(weight_asymm_v6_chamfer_with_stroke_masks=1.0))
    graphed.ENABLED = False
    try:
        want, _, _ = _compute(ref, outs, data)
    finally:
        graphed.ENABLED = True
    assert _rel(call(1.0), want) < 1e-6             # the survivor still computes the loss
    # another batch size, seen once: eager, nothing recorded for it
    small = {k: (v[:2] if isinstance(v, torch.Tensor) and v.dim() and v.shape[0] == B else v) for k, v in data.items()}
    n = len(lh._graph_runners)
    loss, _, _ = _compute(lh, [None if o is None else o[:2] for o in outs], small)
    assert not type(loss.grad_fn).__name__.startswith("_LossReplay") and len(lh._graph_runners) == n


def test_hooks_keep_the_model_eager(mods):
    """A forward hook on a submodule or a gradient hook on a parameter observes the eager code; a replay would silence it."""
    graphed, pc, pu, synthetic = mods
    x = _clouds(4, 1024, 70)
    m = _model(pc, synthetic, seed=9)
    fired = []
    h = m.sa1.register_forward_hook(lambda mod, a, out: fired.append(1))
    for _ in range(graphed.WARM + 2):
        m.zero_grad()
        o = m(x)[0]
        assert not type(o.grad_fn).__name__.startswith("_Replay")
        o.sum().backward()
    assert len(fired) == graphed.WARM + 2 and not m.__dict__.get("_graph_runners")
    h.remove()
    seen = []
    h = m.fc3.weight.register_hook(lambda g: seen.append(float(g.abs().max())))
    for _ in range(graphed.WARM + 2):
        m.zero_grad()
        o = m(x)[0]
        assert not type(o.grad_fn).__name__.startswith("_Replay")
        o.sum().backward()
    assert len(seen) == graphed.WARM + 2
    h.remove()
    for i in range(graphed.WARM + 1):               # hooks gone: recorded after the usual warm-up
        m.zero_grad()
        o = m(x)[0]
        o.sum().backward()
    assert type(o.grad_fn).__name__.startswith("_Replay")


def test_buffers_stay_reserved_while_the_autograd_graph_lives(mods):
    """`l1 = f(b1); l2 = f(b2); (l1 + l2).backward()` with f dropping the model's outputs: the first call's autograd graph outlives its output
    tensors, so the second call must not reuse the recorded buffers (it did, and l1's backward raised)."""
    graphed, pc, pu, synthetic = mods
    from maskplanner_amd.loss_handler import LossHandler, maskplanner_loss_config
    cat = synthetic.CATEGORIES["cuboids"]
    B, N = 4, 1024
    m = _model(pc, synthetic, seed=10)
    lh = LossHandler(["asymm_v6_chamfer_with_stroke_masks"], maskplanner_loss_config())
    batches = [synthetic.make_batch(320 + i, B, N, _narrow(synthetic, 2830, 2940), "cuboid") for i in range(2)]   # (two widths, one capacity bucket)
    assert batches[0]["traj_as_pc"].shape != batches[1]["traj_as_pc"].shape

    def f(b):
        out = m(b["point_cloud"].permute(0, 2, 1).cuda().float())
        return lh.compute(y_pred=out[0], y=b["traj"].cuda().float(), pred_stroke_masks=out[1], mask_scores=out[2], seg_logits=out[3],
                          stroke_ids=b["stroke_ids"], traj_as_pc=b["traj_as_pc"])[0]
    for i in range(graphed.WARM + 2):
        m.zero_grad()
        f(batches[i % 2]).backward()
    assert any(r.graph_r is not None for r in m._graph_runners.values()) and any(r.graph_l is not None for r in lh._graph_runners.values())
    m.zero_grad()
    l1 = f(batches[0])
    l2 = f(batches[1])                              # l1's graph is alive: this one runs eagerly, model and loss
    assert type(l1.grad_fn).__name__.startswith("_LossReplay") and not type(l2.grad_fn).__name__.startswith("_LossReplay")
    (l1 + l2).backward()
    both = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    assert all(torch.isfinite(g).all() for g in both.values())
    # and once that graph is gone the recorded path is taken again
    del l1, l2
    m.zero_grad()
    l3 = f(batches[0])
    assert type(l3.grad_fn).__name__.startswith("_LossReplay")
    l3.backward()


def test_replaced_parameters_drop_the_recording(mods):
    graphed, pc, pu, synthetic = mods
    m = _model(pc, synthetic, seed=8)
    x = _clouds(4, 1024, 60)
    for _ in range(graphed.WARM + 1):
        m.zero_grad()
        m(x)[0].sum().backward()
    assert any(r.graph_r is not None for r in m._graph_runners.values())
    with torch.no_grad():
        m.fc3.weight.data = m.fc3.weight.data.clone()          # the graphs still read the old storage
    m.zero_grad()
    o = m(x)[0]
    assert not type(o.grad_fn).__name__.startswith("_Replay") and not m.__dict__.get("_graph_runners")
    o.sum().backward()


def _g5_model(pc):
    return pc.PointNet2Regressor_StrokeMasks(outdim=12, outdim_orient=12, weight_orient=0.25, out_vectors=99, hidden_size=(64, 64),
                                             pred_stroke_masks=True, n_stroke_masks=6, mask_confidence_scores=True, segment_confidence_scores=False)


def _load(module, g, prefix):
    module.load_state_dict({k[len(prefix):]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(prefix)}, strict=True)
    return module.cuda()


def _close(a, b, what, rtol=1e-5, atol=1e-5):
    a, b = a.detach().cpu().numpy(), np.asarray(b)
    assert a.shape == b.shape, what
    err, scale = np.abs(a - b).max(), max(np.abs(b).max(), 1.0)
    assert err <= atol + rtol * scale, f"{what}: max err {err:.3e} (scale {scale:.3e})"


def test_replayed_forward_against_the_reference_fixtures(mods, golden):
    """The recorded path pinned to the reference itself: g5 (the reference model's eval-mode outputs, 1e-5) through the forward-only graph, and
    g15 (train mode, dropout 0: outputs, running statistics, gradients) through the forward + backward graphs -- the model called until it
    replays, the FPS starts of the fixtures supplied per call as the loop's draws would be."""
    graphed, pc, pu, synthetic = mods
    g5, g15 = golden("g5_model"), golden("g15_train")
    x5 = torch.from_numpy(np.ascontiguousarray(g5["xyz"])).cuda().permute(0, 2, 1)
    model = _load(_g5_model(pc), g5, "sd_").eval()
    with torch.no_grad():
        for i in range(graphed.WARM + 2):
            with pu.fps_start_override([g5["fps_start1"], g5["fps_start2"]]):
                out, sm_out, mask_conf, seg_conf = model(x5)
    assert any(r.graph_f is not None for r in model._graph_runners.values()) and seg_conf is None
    _close(out, g5["out"], "out"); _close(sm_out, g5["sm_out"], "sm_out"); _close(mask_conf, g5["mask_conf"], "mask_conf")
    # train mode: every call starts from the fixture's state (the running statistics and counters it left are compared after ONE pass)
    x15 = torch.from_numpy(np.ascontiguousarray(g15["xyz"])).cuda().permute(0, 2, 1)
    model = _load(_g5_model(pc), g5, "sd_").train()
    model.dropout.p = 0.0
    state = {k: v.clone() for k, v in model.state_dict().items()}
    w_out, w_sm = torch.from_numpy(g15["w_out"]).cuda(), torch.from_numpy(g15["w_sm"]).cuda()
    for i in range(graphed.WARM + 2):
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(state[k])
        model.zero_grad()
        with pu.fps_start_override([g15["fps_start1"], g15["fps_start2"]]):
            out, sm_out, mask_conf, _ = model(x15)
        ((out * w_out).sum() + (sm_out * w_sm).sum() + mask_conf.sum()).backward()
    assert type(out.grad_fn).__name__.startswith("_Replay")
    _close(out, g15["out"], "out", rtol=1e-4); _close(sm_out, g15["sm_out"], "sm_out", rtol=1e-4); _close(mask_conf, g15["mask_conf"], "mask_conf", rtol=1e-4)
    for k, v in model.state_dict().items():
        if "running" in k:
            _close(v, g15["after_" + k], k, rtol=2e-5, atol=1e-6)
        elif "num_batches" in k:
            assert int(v) == int(g15["after_" + k]), k
    params = dict(model.named_parameters())
    for k in g15.files:
        if k.startswith("grad_"):
            got, want = params[k[5:]].grad.cpu(), torch.from_numpy(g15[k])
            assert float((got - want).norm() / want.norm()) < 1e-2, k


@pytest.mark.parametrize("tag", ["cub", "win"])
def test_replayed_loss_against_the_reference_fixture(mods, golden, tag):
    """g7 (the reference's asymm_v6 loss with stroke masks: value and the gradients of its three differentiable inputs) through the loss's
    recorded graphs: compute() called until it replays, fresh leaves per call as the loop's model outputs are."""
    graphed, pc, pu, synthetic = mods
    from maskplanner_amd.loss_handler import LossHandler, maskplanner_loss_config
    g = golden("g7_mask")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    lh = LossHandler(["asymm_v6_chamfer_with_stroke_masks"], maskplanner_loss_config(explicit_no_stroke_weight=float(g[tag + "_no_stroke_weight"])))
    for i in range(graphed.WARM + 2):
        yp, mk, sc = (dev(g[tag + k]).requires_grad_(True) for k in ("_y_pred", "_masks", "_scores"))
        loss, terms = lh.compute(y_pred=yp, y=dev(g[tag + "_traj"]), pred_stroke_masks=mk, mask_scores=sc, seg_logits=None,
                                 stroke_ids=torch.from_numpy(g[tag + "_stroke_ids"]), traj_as_pc=torch.from_numpy(g[tag + "_traj_as_pc"]))
        loss.backward()
    assert type(loss.grad_fn).__name__.startswith("_LossReplay") and isinstance(terms, np.ndarray) and terms.shape == (1,)
    _close(loss, g[tag + "_loss"], "loss", rtol=1e-5)
    _close(torch.from_numpy(terms), g[tag + "_loss"].reshape(1), "terms", rtol=1e-5)
    _close(yp.grad, g[tag + "_g_y_pred"], "g_y_pred", rtol=1e-4, atol=1e-6)
    _close(mk.grad, g[tag + "_g_masks"], "g_masks", rtol=1e-5, atol=1e-6)
    _close(sc.grad, g[tag + "_g_scores"], "g_scores", rtol=1e-5, atol=1e-6)


def test_recordings_keep_the_garbage_collector_out(mods):
    """[r6] A dead reference cycle that owns a pinned host buffer must not be collected WHILE a stream records (its release queries events:
    illegal under capture, raised inside a destructor -- the process aborted in a test session of this round).  harness.recording collects
    before the capture and disables the collector inside it; here such garbage exists when a training step is recorded, and the collector's
    state is restored afterwards."""
    import gc
    graphed, pc, pu, synthetic = mods
    from maskplanner_amd.harness import TrainStep, recording

    class Holder:
        pass
    was = gc.isenabled()
    gc.disable()
    try:
        for _ in range(8):                       # cycles with pinned buffers that have been the source of asynchronous copies
            a, b = Holder(), Holder()
            a.other, b.other = b, a
            a.buf = torch.empty(1 << 16, dtype=torch.float32).pin_memory()
            a.dev = torch.empty(1 << 16, device="cuda")
            a.dev.copy_(a.buf, non_blocking=True)
            del a, b
    finally:
        if was:
            gc.enable()
    ts = TrainStep("cuboids", B=2, N=1024, seed=5, graph=True, hidden_size=(64, 64))
    losses = [float(ts.step()) for _ in range(6)]
    assert ts._graph is not None and np.isfinite(losses).all()
    assert gc.isenabled() == was
    g = torch.cuda.CUDAGraph()
    x = torch.zeros(8, device="cuda")
    with recording(g):
        assert not gc.isenabled()
        y = x + 1
    assert gc.isenabled() == was
    g.replay()
    torch.cuda.synchronize()
    assert float(y.sum()) == 8.0
