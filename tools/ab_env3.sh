#!/bin/bash
# alternate several environments of the working tree on one box: ab_env3.sh "ENV_A" "ENV_B" "ENV_C" ... (bench medians, two rounds)
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  for E in "$@"; do
    env $E python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
print('[$E]', 'mean', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median', 0),3), 'min', round(d.get('step_ms_min', 0),3))"
  done
done
