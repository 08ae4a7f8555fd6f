#!/bin/bash
# config 5: kernel trace of the replayed step, all kernels of one step on the chain with durations (gpurun_out/<tag>/seq.txt)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-c5t}; rm -rf $O; mkdir -p $O
A="--encoder msg --category containers --points 10240 --dtype bf16 --no-cpu-baseline --no-side-legs"
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 bench.py $A --steps 8 --warmup 3 > $O/tr.log 2>&1
python3 tools/step_gaps.py $O/tr 3
find $O/tr -type f ! -name '*kernel_trace.csv' -delete
python3 bench.py $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
print('config5 mean', round(d['ms_per_step'],3), 'median', round(d['step_ms_median'],3), 'value', round(d['value'],1))"
