# A/B of the role-split fused backward (MP_BF_ROLES bit mask: 1 = 256-output layer, 2 = 128 -> 128) on one box, alternating; tests first
cd $GRAFT_REPO_ROOT
MP_BF_ROLES=3 python -m pytest tests/test_gpu_split.py tests/test_gpu_modules.py -q -x 2>&1 | tail -2
for i in 1 2 3; do for v in 0 1 2 3; do
  echo -n "roles=$v: "; MP_BF_ROLES=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), {n[:28]: round(v,1) for n,v in k.items() if 'bwd_f' in n})"
done; done
