# alternate the working tree and ab_prev/ (tools/ab_setup.sh) on the same box: ms per step of each, three rounds
# AB_ARGS: extra bench.py flags for the working tree only (e.g. AB_ARGS=--no-graph to compare launch modes of one tree)
for i in 1 2 3; do
for v in cur prev; do
  if [ $v = prev ]; then d=$GRAFT_REPO_ROOT/ab_prev; extra=""; else d=$GRAFT_REPO_ROOT; extra="$AB_ARGS"; fi
  (cd $d && python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs $extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
print('$v', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median', 0),3), 'min', round(d.get('step_ms_min', 0),3))
if $i == 3:
    k=d['kernels_us_per_step']; print('   ', {n: v for n, v in list(k.items())[:22]})
")
done; done
