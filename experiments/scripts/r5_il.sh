# [r5] chunk-interleaved workgroups in the fused backward (MP_BF_IL): parity subset, then same-box A/B
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_split.py tests/test_gpu_modules.py tests/test_gpu_bnsites.py tests/test_gpu_routing.py tests/test_gpu_bf16.py -q -x 2>&1 | tail -3 | cut -c1-300
for i in 1 2 3 4; do for v in 0 1; do
  echo -n "[${AB:-MP_BF_IL}=$v]: "; env ${AB:-MP_BF_IL}=$v timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys,re
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
sel={n: v for n,v in k.items() if re.search('bwd_|fwd_chunk', n)}
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), 'sum', round(sum(sel.values()),1), {n[:32]: round(v,1) for n,v in sorted(sel.items(), key=lambda kv: -kv[1])[:9]})"
done; done
