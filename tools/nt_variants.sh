# A/B: the 128 -> 128 fused backward with four waves per workgroup (MP_BF_NT256=2) against eight; both with MP_PD2=3
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
OBJS=$(for f in *.hip; do [ $f != sa_mlp.hip ] && echo $GRAFT_REPO_ROOT/maskplanner_amd/lib/obj/${f%.hip}.o; done)
VARS="${VARS:-0 2}"
for v in $VARS; do
  d=/tmp/nt$v; mkdir -p $d
  hipcc -DMP_BF_NT256=$v ${EXTRA} -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c sa_mlp.hip -o $d/sa_mlp.o 2>/dev/null &
done
wait
for v in $VARS; do d=/tmp/nt$v; hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $d/lib.so $d/sa_mlp.o $OBJS; done
cd $GRAFT_REPO_ROOT
last=$(echo $VARS | awk '{print $NF}')
MASKPLANNER_HIP_LIB=/tmp/nt$last/lib.so python -m pytest tests/test_gpu_split.py tests/test_gpu_modules.py -q -x 2>&1 | tail -2
for i in 1 2 3; do for v in $VARS; do
  echo -n "nt256=$v: "; MASKPLANNER_HIP_LIB=/tmp/nt$v/lib.so python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), {n[:28]: round(v,1) for n,v in k.items() if 'bwd_f' in n})"
done; done
