for i in 1 2; do for v in 512 256 1024 2048; do
  echo -n "MP_FWD_WGS=$v: "; MP_FWD_WGS=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), {n[17:40]: round(v,1) for n,v in k.items() if 'fwd_chunk' in n})"
done; done
