# timing builds of the fused backward kernels: lane mapping of the plane writes (MP_MAPWIDE) on one box, alternating
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
for v in 0 1; do
  d=/tmp/bw$v; mkdir -p $d
  hipcc -DMP_MAPWIDE=$v -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c sa_mlp.hip -o $d/sa_mlp.o
  hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $d/lib.so $d/sa_mlp.o $(ls ../lib/obj/*.o | grep -v sa_mlp.o)
done
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for v in 0 1; do
  echo -n "mapwide=$v: "; MASKPLANNER_HIP_LIB=/tmp/bw$v/lib.so python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), {n[:28]: round(v,1) for n,v in k.items() if 'bwd_f' in n})"
done; done
