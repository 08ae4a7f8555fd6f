# timing builds of the factorised first-layer kernels: rows in flight (MP_FACT_U) and positions per workgroup (MP_FACT_PPB), config 5, one box
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
for v in "4 256" "8 256" "4 512" "8 512" "8 1024" "2 256"; do
  set -- $v; d=/tmp/fa$1_$2; mkdir -p $d
  hipcc -DMP_FACT_U=$1 -DMP_FACT_PPB=$2 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -I../../include -c sa_mlp.hip -o $d/sa_mlp.o || continue
  hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $d/lib.so $d/sa_mlp.o $(ls ../lib/obj/*.o | grep -v sa_mlp.o)
done
cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in "4 256" "8 256" "4 512" "8 512" "8 1024" "2 256"; do
  set -- $v; [ -f /tmp/fa$1_$2/lib.so ] || continue
  echo -n "U=$1 ppb=$2: "; MASKPLANNER_HIP_LIB=/tmp/fa$1_$2/lib.so python bench.py --category containers --points 10240 --encoder msg --dtype f32 --steps 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); k=d['kernels_us_per_step']
print(round(d['ms_per_step'],3), {n[:26]: round(v,1) for n,v in k.items() if 'factored' in n})"
done; done
