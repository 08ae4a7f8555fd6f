# column-slab width of the grouping backward's gather-reduce (MP_GROUP_SLAB), config 5 (D = 320) and the default bench (D = 128), one box
for i in 1 2; do for v in 0 64; do
  echo -n "c5 slab=$v: "; MP_GROUP_SLAB=$v python bench.py --category containers --points 10240 --encoder msg --dtype f32 --steps 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); k=d['kernels_us_per_step']
print(round(d['ms_per_step'],3), {n[:26]: round(v,1) for n,v in k.items() if 'group_bwd' in n})"
done; done
for i in 1 2 3; do for v in 0 64; do
  echo -n "default slab=$v: "; MP_GROUP_SLAB=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); k=d['kernels_us_per_step']
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), {n[:26]: round(v,1) for n,v in k.items() if 'group_bwd' in n})"
done; done
