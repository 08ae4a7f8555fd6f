#!/bin/bash
# alternate two environments of the working tree on one box: ab_env2.sh "ENV_A" "ENV_B" [bench args]
cd "$GRAFT_REPO_ROOT"
A="$1"; B="$2"; shift 2
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    env $E python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
print('$v [$E]', 'mean', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median', 0),3), 'min', round(d.get('step_ms_min', 0),3))"
  done
done
