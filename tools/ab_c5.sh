# config 5 (containers, N = 10240, MSG encoder, bf16): working tree against ab_prev/ (tools/ab_setup.sh) on one box, three alternations
A="--encoder msg --category containers --points 10240 --dtype bf16 --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs"
for i in 1 2 3; do
for v in cur prev; do
  if [ $v = prev ]; then d=$GRAFT_REPO_ROOT/ab_prev; else d=$GRAFT_REPO_ROOT; fi
  (cd $d && python bench.py $A 2>/dev/null | FILTER="$FILTER" python -c "
import json,sys,re,os
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
sel={n[:40]: round(v,1) for n,v in k.items() if re.search(os.environ.get('FILTER') or 'factored', n)}
print('$v', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median', 0),3), 'min', round(d.get('step_ms_min', 0),3), sel)
")
done; done
