# timing builds of the fused backward of the 256-output layer: K split of dX on / off, lane mapping of the dZ plane writes
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
for v in "0 0" "1 0" "0 1" "1 1"; do
  set -- $v; d=/tmp/bv$1$2; mkdir -p $d
  hipcc -DMP_BWD_KSPLIT=$1 -DMP_MAP256=$2 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c sa_mlp.hip -o $d/sa_mlp.o
  hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $d/lib.so $d/sa_mlp.o $(ls ../lib/obj/*.o | grep -v sa_mlp.o)
done
cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in 00 10 01 11; do
  echo -n "ksplit/map256=$v: "; MASKPLANNER_HIP_LIB=/tmp/bv$v/lib.so python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), {n: round(v,1) for n,v in k.items() if 'bwd_fused' in n})"
done; done
MASKPLANNER_HIP_LIB=/tmp/bv10/lib.so MP_SA_SPLIT=1 python tools/split_check.py 2>&1 | grep -E "level|input"
