#!/bin/bash
# usage: r4_check.sh <out tag> [pytest -k expression]: GPU tests (all or selected), step probe, kernel trace + gap analysis, bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; rm -rf $O; mkdir -p $O
if [ -n "$2" ]; then K=(-k "$2"); else K=(); fi
timeout 2400 python3 -m pytest tests -x -q -m gpu "${K[@]}" > $O/tests.txt 2>&1
python3 tools/boundary_probe.py 60 > $O/probe.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-legs > $O/tr.log 2>&1
python3 tools/step_sequence.py $O/tr 3 > $O/seq.txt 2>&1
python3 tools/step_gaps.py $O/tr 6 > $O/gaps.txt 2>&1
find $O/tr -type f ! -name '*kernel_trace.csv' -delete
python3 bench.py --no-cpu-baseline --no-side-legs > $O/bench.json 2> $O/bench.err
tail -4 $O/tests.txt; grep "device step" $O/probe.txt; cat $O/gaps.txt; python3 -c "
import json; d=json.load(open('$O/bench.json')); print('bench mean', d['ms_per_step'], 'median', d['step_ms_median'], 'min', d['step_ms_min'], 'value', d['value'])"
