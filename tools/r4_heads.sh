#!/bin/bash
# head-block kernels: tests, then a kernel trace of the step's head section
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_gpu_heads.py -x -q 2>&1 | tail -15
rm -rf gpurun_out/heads_tr
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/heads_tr -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-legs > gpurun_out/heads_tr.log 2>&1
python3 tools/step_gaps.py gpurun_out/heads_tr 4
