# timing builds of the factorised first layer's sorted-row reduce: rows in flight (MP_FACT_RU, compile time) x rows per wave (MP_FACT_CHUNK), one box
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
for u in 4 8; do
  d=/tmp/fr$u; mkdir -p $d
  hipcc -DMP_FACT_RU=$u -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -I../../include -c sa_mlp.hip -o $d/sa_mlp.o || continue
  hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $d/lib.so $d/sa_mlp.o $(ls ../lib/obj/*.o | grep -v sa_mlp.o)
done
cd $GRAFT_REPO_ROOT
for i in 1 2; do for u in 4 8; do for c in 32 64 128 256; do
  echo -n "U=$u chunk=$c: "; MP_FACT_CHUNK=$c MASKPLANNER_FACTORED_FIRST=1 MASKPLANNER_HIP_LIB=/tmp/fr$u/lib.so python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); k=d['kernels_us_per_step']
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), {n[:30]: round(v,1) for n,v in k.items() if 'factored' in n})"
done; done; done
