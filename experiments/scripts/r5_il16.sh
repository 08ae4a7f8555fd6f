# [r5] interleaved workgroups in the bf16 ring kernels (MP_S16_IL backward, MP_S16_FIL forward): config-5 parity tests, then same-box A/B
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_split.py -q -x 2>&1 | tail -3 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_modules.py -q -x -k "config5 or steady" 2>&1 | tail -2 | cut -c1-300
for i in 1 2 3; do for v in "MP_S16_IL=0 MP_S16_FIL=0" "MP_S16_IL=1 MP_S16_FIL=0" "MP_S16_IL=1 MP_S16_FIL=1"; do
  echo -n "[$v]: "; env $v timeout 600 python bench.py --category containers --points 10240 --encoder msg --dtype bf16 --steps 40 --warmup 8 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys,re
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
sel={n: v for n,v in k.items() if re.search('stream16', n)}
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), 'sum', round(sum(sel.values()),1), {n[:36]: round(v,1) for n,v in sorted(sel.items(), key=lambda kv: -kv[1])[:8]})"
done; done
