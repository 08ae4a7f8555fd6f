# [r5] fp32 ring kernel for the 256-output layer (MP_S32): parity subset, then same-box A/B against bwd_roles_kernel
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_modules.py tests/test_gpu_bnsites.py tests/test_gpu_routing.py tests/test_gpu_ops.py -q -x 2>&1 | tail -3
for i in 1 2 3; do for v in 0 1; do
  echo -n "[MP_S32=$v]: "; MP_S32=$v timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys,re
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
sel={n: v for n,v in k.items() if re.search('bwd_', n)}
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), 'sum', round(sum(sel.values()),1), {n[:40]: round(v,1) for n,v in sorted(sel.items(), key=lambda kv: -kv[1])[:6]})"
done; done
