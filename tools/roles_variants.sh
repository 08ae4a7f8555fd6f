# compile-time variants of the role-split fused backward (bwd_roles_kernel) on one box: EXTRA flag sets in VARS, run with MP_BF_ROLES=1
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
OBJS=$(for f in *.hip; do [ $f != sa_mlp.hip ] && echo $GRAFT_REPO_ROOT/maskplanner_amd/lib/obj/${f%.hip}.o; done)
IFS=';' read -ra VV <<< "${VARS:--DMP_ROLES_SPLITSTAGE=0;-DMP_ROLES_SPLITSTAGE=1}"
n=0
for v in "${VV[@]}"; do
  d=/tmp/rv$n; mkdir -p $d
  hipcc $v -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c sa_mlp.hip -o $d/sa_mlp.o 2>/dev/null &
  n=$((n+1))
done
wait
n=0; for v in "${VV[@]}"; do d=/tmp/rv$n; hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $d/lib.so $d/sa_mlp.o $OBJS; n=$((n+1)); done
cd $GRAFT_REPO_ROOT
last=$((n-1))
MP_BF_ROLES=1 MASKPLANNER_HIP_LIB=/tmp/rv$last/lib.so python -m pytest tests/test_gpu_split.py tests/test_gpu_modules.py -q -x 2>&1 | tail -2
for i in 1 2 3; do n=0; for v in "${VV[@]}"; do
  echo -n "[$v]: "; MP_BF_ROLES=1 MASKPLANNER_HIP_LIB=/tmp/rv$n/lib.so python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), {n[:28]: round(v,1) for n,v in k.items() if 'bwd_fused_kernel<3, 256' in n})"
  n=$((n+1))
done; done
