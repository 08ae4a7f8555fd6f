#!/bin/bash
# usage: r4_trace.sh <tag>: kernel trace of the replayed step (gap analysis) + one bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-legs > $O/tr.log 2>&1
python3 tools/step_gaps.py $O/tr 4
find $O/tr -type f ! -name '*kernel_trace.csv' -delete
python3 bench.py --no-cpu-baseline --no-side-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
print('bench mean', round(d['ms_per_step'],3), 'median', round(d['step_ms_median'],3), 'min', round(d['step_ms_min'],3), 'value', round(d['value'],1))"
