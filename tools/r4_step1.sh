#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s1; rm -rf $O; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "pipelin or streamed or launch_mode or graph or two_graph" > $O/tests.txt 2>&1
python3 tools/boundary_probe.py 60 > $O/probe.txt 2>&1
python3 tools/torch_ops_order.py > $O/ops_order.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-legs > $O/tr.log 2>&1
python3 tools/step_sequence.py $O/tr 3 > $O/seq.txt 2>&1
find $O/tr -type f ! -name '*kernel_trace.csv' -delete
python3 bench.py --no-cpu-baseline --no-side-legs > $O/bench.json 2> $O/bench.err
tail -5 $O/tests.txt; tail -8 $O/probe.txt; tail -1 $O/seq.txt; cut -c1-300 $O/bench.json
