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
st = ts._stream
real = st.collate_next
def no_host():
    return st.stage["point_cloud"], st.stage_starts
st.collate_next = no_host
out["streamed_without_collation"] = run(ts)
st.collate_next = real
# host pieces of one collation, alone
t0 = time.perf_counter()
for _ in range(20):
    st.collate_next()
torch.cuda.synchronize()
out["collate_next_alone_ms"] = (time.perf_counter() - t0) / 20 * 1e3
print(json.dumps(out))
