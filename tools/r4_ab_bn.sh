#!/bin/bash
# consumer-side BatchNorm finalize (MASKPLANNER_BN_FUSED=1) against the finalize launches (=0) on one box: per-kernel averages without the
# side streams (clean kernel times), then the default step, alternating
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/abbn; rm -rf $O; mkdir -p $O
for v in 0 1; do
  MASKPLANNER_OVERLAP_SAMPLING=0 MASKPLANNER_SPLIT_ADAM=0 MASKPLANNER_BN_FUSED=$v rocprofv3 --kernel-trace --output-format csv -d $O/tr$v -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-legs > $O/tr$v.log 2>&1
  find $O/tr$v -type f ! -name '*kernel_trace.csv' -delete
done
for v in 0 1 0 1 0 1; do
  MASKPLANNER_BN_FUSED=$v python3 bench.py --steps 40 --no-cpu-baseline --no-side-legs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('default step, fused=$v', 'median', round(d['step_ms_median'],4), 'min', round(d['step_ms_min'],4))"
done
