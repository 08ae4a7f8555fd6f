# [r5] in-graph kernel marks: do event-record nodes time correctly, and what do they cost?  A/B of the step with and without marks
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for m in "bwd_roles_kernel<3" "none_matches"; do
  echo -n "[marks=$m]: "; MASKPLANNER_BENCH_MARKS="$m" timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
r=d['roofline']
print(round(d['value']), round(d['ms_per_step'],4), round(d['step_ms_median'],4), r['kernel'], round(r['avg_us'],1), round(r.get('eager_step_avg_us',0),1), r.get('timed_by','')[:40])"
done; done
echo "--- steps 20 warmup 5 (the driver's invocation)"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
print(round(d['value']), round(d['ms_per_step'],4), round(d['step_ms_median'],4), d['roofline']['kernel'], round(d['roofline']['avg_us'],1))"
