cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5k
cat > /tmp/chk.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from maskplanner_amd.harness import TrainStep
def run(ts, n):
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for i in range(n):
        marks[i].record()
        ts.step()
    marks[-1].record()
    torch.cuda.synchronize()
    t = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(n))
    return round(t[n // 2], 3)
out = []
for kind in sys.argv[1].split(","):
    ts = TrainStep("cuboids", B=32, N=5120, stream_batches=4 if kind == "s" else 0)
    while ts._graph is None:
        ts.step()
    for _ in range(8):
        ts.step()
    torch.cuda.synchronize()
    out.append((kind, run(ts, 40)))
    del ts
    torch.cuda.empty_cache()
print(out)
PY
for o in "s,r" "r,s" "s"; do python3 /tmp/chk.py $o 2>/dev/null; done > gpurun_out/r5k/streams.txt
python -m pytest tests/test_gpu_modules.py -m gpu -x -q -k "stream or plan_carried or collat" 2>&1 | tail -2 >> gpurun_out/r5k/streams.txt
cat gpurun_out/r5k/streams.txt
