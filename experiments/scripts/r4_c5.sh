#!/bin/bash
# config 5 (containers, N = 10240, MSG encoder, bf16 grouped MLP): bench line + per-kernel stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-c5}; rm -rf $O; mkdir -p $O
A="--encoder msg --category containers --points 10240 --dtype bf16 --no-cpu-baseline --no-side-legs"
python3 bench.py $A > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 bench.py $A --steps 8 --warmup 3 > $O/tr.log 2>&1
python3 - <<'PY' > $O/stats.txt
import csv,glob,collections,sys,os
O=os.environ.get("O","")
PY
f=$(ls $O/tr/*/*kernel_stats.csv | head -1); head -40 $f | cut -d, -f1-4 > $O/kernel_stats_top.csv
find $O/tr -type f ! -name '*kernel_stats.csv' -delete
python3 -c "
import json; d=json.load(open('$O/bench.json')); print('config5 mean', d['ms_per_step'], 'median', d['step_ms_median'], 'value', d['value']); print(d['roofline']['kernel'], d['roofline']['avg_us'])"
head -30 $O/kernel_stats_top.csv
