# the few-row (group_all) level with its operands pre-split into bf16 planes (PREC 4, MP_PLANES=1) against the split-in-the-tile kernels, one box
cd $GRAFT_REPO_ROOT
MP_PLANES=1 python -m pytest tests/test_gpu_split.py tests/test_gpu_modules.py tests/test_gpu_ops.py -q -x 2>&1 | tail -2
for i in 1 2 3; do for v in 0 1; do
  echo -n "planes=$v: "; MP_PLANES=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
sel={n: v for n,v in k.items() if any(t in n for t in ('pos_gemm','dw_gemm','act_split','w_split'))}
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), 'sum', round(sum(sel.values()),1), {n[:40]: round(v,1) for n,v in sel.items()})"
done; done
