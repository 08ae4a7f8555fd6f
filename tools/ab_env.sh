# alternate one environment switch on the same box: AB_VAR=NAME bash tools/ab_env.sh  (values 1 / 0), three rounds
for i in 1 2 3; do
for v in 1 0; do
  env $AB_VAR=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs $AB_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
print('$AB_VAR=$v', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median', 0),3), 'min', round(d.get('step_ms_min', 0),3))
"
done; done
