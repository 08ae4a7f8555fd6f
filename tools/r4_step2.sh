#!/bin/bash
# launch-structure variants of the replayed step on one box: plan at start / mid, head optimizer on its own stream / in line, backward split
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s2; rm -rf $O; mkdir -p $O
run() {  # name, env...
  n=$1; shift
  for r in 1 2; do
    env "$@" python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-side-legs 2> $O/$n.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$n', 'run$r', 'median', round(d['step_ms_median'],4), 'min', round(d['step_ms_min'],4), 'mean', round(d['ms_per_step'],4))"
  done
}
run start_side
run mid_side MASKPLANNER_PLAN_AT=mid
run start_inline MASKPLANNER_SPLIT_ADAM=0
run mid_inline MASKPLANNER_PLAN_AT=mid MASKPLANNER_SPLIT_ADAM=0
run start_3g MASKPLANNER_SPLIT_BACKWARD=1
run mid_3g MASKPLANNER_PLAN_AT=mid MASKPLANNER_SPLIT_BACKWARD=1
run start_side
for v in mid_side mid_inline mid_3g; do
  case $v in
    mid_side) E="MASKPLANNER_PLAN_AT=mid";;
    mid_inline) E="MASKPLANNER_PLAN_AT=mid MASKPLANNER_SPLIT_ADAM=0";;
    mid_3g) E="MASKPLANNER_PLAN_AT=mid MASKPLANNER_SPLIT_BACKWARD=1";;
  esac
  for kv in $E; do export $kv; done
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_$v -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-legs > $O/tr_$v.log 2>&1
  unset MASKPLANNER_PLAN_AT MASKPLANNER_SPLIT_ADAM MASKPLANNER_SPLIT_BACKWARD
  python3 tools/step_sequence.py $O/tr_$v 3 > $O/seq_$v.txt 2>&1
  find $O/tr_$v -type f ! -name '*kernel_trace.csv' -delete
done
timeout 600 python3 -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "pipelin or launch_mode or graph or two_graph" > $O/tests_start.txt 2>&1
MASKPLANNER_PLAN_AT=mid timeout 600 python3 -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "pipelin or launch_mode or graph or two_graph" > $O/tests_mid.txt 2>&1
tail -2 $O/tests_start.txt $O/tests_mid.txt
