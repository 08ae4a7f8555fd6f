cd $GRAFT_REPO_ROOT
for f in 1 0 1 0; do echo "FORK=$f"; MASKPLANNER_DROPIN_GRAPH_FORK=$f python - <<'PY' 2>&1 | grep -v "Warn\|warn\|amdgpu.ids"
import sys, time, torch
sys.path.insert(0, '.')
from maskplanner_amd.harness import DropInLoop
loop = DropInLoop("cuboids", B=32, N=5120)
for _ in range(10): loop.step()
per=[]
for _ in range(40):
    torch.cuda.synchronize(); t0=time.perf_counter(); loop.step(); torch.cuda.synchronize(); per.append((time.perf_counter()-t0)*1e3)
per.sort(); print("median %.2f ms" % per[20], [ (k[0], r.graph_r is not None, r.failed) for k, r in loop.model._graph_runners.items()])
PY
done
