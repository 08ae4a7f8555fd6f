# the role-split fused backward with its raw chunks loaded straight into LDS (MP_BF_ROLES_LDS = 2 | 3 raw stages) against the register-staged one
cd $GRAFT_REPO_ROOT
for v in 2 3; do MP_BF_ROLES_LDS=$v python -m pytest tests/test_gpu_split.py tests/test_gpu_modules.py -q -x 2>&1 | tail -2; done
for i in 1 2 3; do for v in 0 2 3; do
  echo -n "roles_lds=$v: "; MP_BF_ROLES_LDS=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), {n[:28]: round(v,1) for n,v in k.items() if 'bwd_fused_kernel<3, 256' in n})"
done; done
