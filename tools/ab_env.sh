#!/bin/bash
# A/B of one environment switch on the headline step, alternating on the same box: tools/ab_env.sh VAR A B [rounds] [extra bench args]
# prints step_ms_median / min and ms_per_step of every run.
VAR=$1; A=$2; B=$3; R=${4:-3}; shift 4
mkdir -p gpurun_out/ab
for i in $(seq 1 $R); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 60 --warmup 10 --no-side-legs --no-cpu-baseline "$@" > gpurun_out/ab/${VAR}_${v}_$i.json 2> gpurun_out/ab/${VAR}_${v}_$i.err
    python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/ab/${VAR}_${v}_$i.json") if l.startswith("{")][-1])
    print("$VAR=$v", "median %.4f min %.4f mean %.4f" % (d["step_ms_median"], d["step_ms_min"], d["ms_per_step"]), "loss", d["final_loss"])
except Exception as e:
    print("$VAR=$v ERR", e, open("gpurun_out/ab/${VAR}_${v}_$i.err").read()[-800:])
PY
  done
done
