# timing builds: cache policy of the raw Z / G buffer stores of the position-stream kernels (MP_STORE_AUX), alternating on one box
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
for v in 0 2 1 17; do
  d=/tmp/sv$v; mkdir -p $d
  hipcc -DMP_STORE_AUX=$v -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c sa_mlp.hip -o $d/sa_mlp.o || continue
  hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $d/lib.so $d/sa_mlp.o $(ls ../lib/obj/*.o | grep -v sa_mlp.o)
done
cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in 0 2 1 17; do
  [ -f /tmp/sv$v/lib.so ] || continue
  echo -n "store_aux=$v: "; MASKPLANNER_HIP_LIB=/tmp/sv$v/lib.so python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), {n[:30]: round(v,1) for n,v in k.items() if ('fwd_chunk' in n or 'bwd_f' in n)})"
done; done
