#!/bin/bash
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE, each in its own run, counters + kernel trace only) for the other BASELINE configs, so that
# `roofline.traffic` of every bench line is a measured number:  tools/profile_configs.sh r03  ->  gpurun_out/<tag>_cfg_<key>_{fetch,write}/
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {   # key, bench flags
  key=$1; shift
  for c in fetch write; do
    C=FETCH_SIZE; [ $c = write ] && C=WRITE_SIZE
    rm -rf gpurun_out/${TAG}_cfg_${key}_$c
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/${TAG}_cfg_${key}_$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-legs "$@" > gpurun_out/${TAG}_cfg_${key}_$c.log 2>&1
    find gpurun_out/${TAG}_cfg_${key}_$c -type f ! -name '*counter_collection.csv' -delete 2>/dev/null
  done
}
run windows --category windows
run shelves --category shelves
run containers_msg_f32 --category containers --points 10240 --encoder msg --dtype f32
run containers_msg_bf16 --category containers --points 10240 --encoder msg --dtype bf16
ls gpurun_out/${TAG}_cfg_*/*/ 2>/dev/null | head -20
