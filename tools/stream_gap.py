"""[r5] Where the streamed-input step's extra time goes (2.47 vs 2.14 ms): median replayed step of
  resident            the headline configuration
  streamed            a fresh host batch per step (TrainStep(stream_batches=4))
  streamed, no host   the same launches with the host-side collation skipped (staging tensors reused): what the DEVICE side of streaming costs
plus the host time per step inside _launch_sampling (numpy stacking, pinned copies, launches).  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd.harness import TrainStep


def run(ts, k=60, probe=None):
    for _ in range(8):
        ts.step()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
    host = []
    orig = ts._launch_sampling

    def timed(*a, **kw):
        t0 = time.perf_counter()
        r = orig(*a, **kw)
        host.append(time.perf_counter() - t0)
        return r
    ts._launch_sampling = timed
    t0 = time.perf_counter()
    for i in range(k):
        marks[i].record()
        ts.step()
    marks[-1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / k * 1e3
    ts._launch_sampling = orig
    per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(k))
    return {"step_ms_median": per[len(per) // 2], "step_ms_min": per[0], "wall_ms_per_step": wall,
            "host_ms_in_launch_sampling": sum(host) / max(len(host), 1) * 1e3}


out = {}
ts = TrainStep("cuboids", B=32, N=5120)
while ts._graph is None:
    ts.step()
out["resident"] = run(ts)
del ts
torch.cuda.empty_cache()
ts = TrainStep("cuboids", B=32, N=5120, stream_batches=4)
while ts._graph is None:
    ts.step()
out["streamed"] = run(ts)
# where the host's time inside _launch_sampling goes (streamed): per-call wall time of its pieces
import collections
acc = collections.defaultdict(list)
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[name].append(time.perf_counter() - t0); return r
    setattr(obj, name, g)
    return f
saved = [(ts._stream, n, wrap(ts._stream, n)) for n in ("upload", "collate_cloud", "collate_targets", "collated")] + [(ts, n, wrap(ts, n)) for n in ("_sample_levels", "_extras", "_target_aux")]
import numpy as _np
_stack = _np.stack
def tstack(*a, **k):
    t0 = time.perf_counter(); r = _stack(*a, **k); acc["np.stack"].append(time.perf_counter() - t0); return r
_np.stack = tstack
run(ts, 40)
_np.stack = _stack
for o, n, f in saved:
    setattr(o, n, f)
out["host_pieces_ms"] = {k: round(sum(v) / len(v) * 1e3, 3) for k, v in acc.items()}
st = ts._stream
real = (st.collate_cloud, st.collate_targets, st.collated, st.prefetch)
st.collate_cloud = lambda: (st.stage["point_cloud"], st.stage_starts)
st.collate_targets = lambda: None
st.collated = lambda *e: None
st.prefetch = lambda: None
out["streamed_without_collation"] = run(ts)
st.collate_cloud, st.collate_targets, st.collated, st.prefetch = real
# host pieces of one collation, alone
t0 = time.perf_counter()
for _ in range(20):
    st.upload()
    st.collate_next()
    st.collated()
torch.cuda.synchronize()
out["upload_and_collate_alone_ms"] = (time.perf_counter() - t0) / 20 * 1e3
print(json.dumps(out))
