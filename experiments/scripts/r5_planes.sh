# [r5] two-plane backward: the suite under the default (2 planes), then same-box A/B of MP_BWD_PLANES=2 / 3 (default bench, config 4 windows), and the per-phase cycle counts
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r5p/tests.txt
for i in 1 2 3; do for v in 2 3; do
  MP_BWD_PLANES=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
print('planes=$v', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median', 0),3), 'min', round(d.get('step_ms_min', 0),3), {n[:40]: round(x,1) for n,x in k.items() if 'bwd_' in n})
"
done; done > gpurun_out/r5p/ab.txt 2>&1
EXTRA= bash tools/roles_timing.sh > gpurun_out/r5p/roles_timing.txt 2>&1
