# generic same-box A/B of compile-time variants of sa_mlp.hip (SRC=sa_stream16.hip: of that file):  VARS="-DA=0;-DA=1" FILTER="pos_gemm|dw_gemm" [ENVV="MP_X=1"] [TESTS="tests/test_gpu_split.py"] bash tools/sa_variants.sh
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
SRC=${SRC:-sa_mlp.hip}
OBJS=$(for f in *.hip; do [ $f != $SRC ] && echo $GRAFT_REPO_ROOT/maskplanner_amd/lib/obj/${f%.hip}.o; done)
IFS=';' read -ra VV <<< "${VARS}"
n=0
for v in "${VV[@]}"; do
  d=/tmp/sv$n; mkdir -p $d
  hipcc $v -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c $SRC -o $d/sa_mlp.o 2>/dev/null &
  n=$((n+1))
done
wait
n=0; for v in "${VV[@]}"; do d=/tmp/sv$n; hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $d/lib.so $d/sa_mlp.o $OBJS; n=$((n+1)); done
cd $GRAFT_REPO_ROOT
last=$((n-1))
env $ENVV MASKPLANNER_HIP_LIB=/tmp/sv$last/lib.so python -m pytest ${TESTS:-tests/test_gpu_split.py tests/test_gpu_modules.py} -q -x 2>&1 | tail -2
for i in 1 2 3; do n=0; for v in "${VV[@]}"; do
  echo -n "[$v]: "; env $ENVV MASKPLANNER_HIP_LIB=/tmp/sv$n/lib.so python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs ${BENCH_ARGS} 2>/dev/null | FILTER="$FILTER" python -c "
import json,sys,re,os
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
sel={n: v for n,v in k.items() if re.search(os.environ.get('FILTER') or '.', n)}
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), 'sum', round(sum(sel.values()),1), {n[:44]: round(v,1) for n,v in sorted(sel.items(), key=lambda kv: -kv[1])[:12]})"
  n=$((n+1))
done; done
