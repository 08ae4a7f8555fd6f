# factorised first layer (MASKPLANNER_FACTORED_FIRST) against the grouped one: default bench, alternating on one box
for i in 1 2 3; do for v in 0 1; do
  echo -n "factored=$v: "; MASKPLANNER_FACTORED_FIRST=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); k=d['kernels_us_per_step']
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), round(d['final_loss'],4), {n[:28]: round(v,1) for n,v in k.items() if any(t in n for t in ('factored','dz_store','group_kernel','group_bwd','bwd_first','0, 4','gathered'))})"
done; done
