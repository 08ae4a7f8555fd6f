#!/bin/bash
# rocprofv3 passes over the default bench (run on the GPU box through gpurun): kernel trace + stats, then the PMC passes, each in
# its own run (counters never together with trace domains other than --kernel-trace; FETCH_SIZE and WRITE_SIZE do not fit one pass).
#   tools/profile_round.sh r02 [extra bench flags]   ->  gpurun_out/<tag>_{bench,fetch,write,mfma}/ ; then tools/make_profile_summary.py <tag>
TAG=${1:-r04}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
COMMON="--no-cpu-baseline --no-side-legs $*"
rm -rf gpurun_out/${TAG}_bench gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_mfma
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_bench -- python3 bench.py --steps 10 --warmup 3 $COMMON > gpurun_out/${TAG}_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_fetch -- python3 bench.py --steps 3 --warmup 1 $COMMON > gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_write -- python3 bench.py --steps 3 --warmup 1 $COMMON > gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/${TAG}_mfma -- python3 bench.py --steps 3 --warmup 1 $COMMON > gpurun_out/${TAG}_mfma.log 2>&1
# keep only the CSVs (the merge back is capped at 64 MiB)
find gpurun_out/${TAG}_bench gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_mfma -type f ! -name '*.csv' -delete 2>/dev/null
find gpurun_out/${TAG}_bench -name '*kernel_trace.csv' -size +20M -delete 2>/dev/null
ls -la gpurun_out/${TAG}_*/*/ 2>/dev/null | head -40
tail -c 400 gpurun_out/${TAG}_bench.log
