# same-box per-kernel comparison of the working tree's library against ab_prev's (tools/ab_setup.sh): rocprofv3 kernel stats of
# a short default bench with each, twice, filtered to the kernels named in $1 (egrep pattern)
PAT=${1:-finalize}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for i in 1 2; do for v in cur prev; do
  if [ $v = prev ]; then export MASKPLANNER_HIP_LIB=$GRAFT_REPO_ROOT/ab_prev/maskplanner_amd/lib/libmaskplanner_hip.so; else unset MASKPLANNER_HIP_LIB; fi
  rm -rf gpurun_out/skab; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/skab -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs > gpurun_out/skab.log 2>&1
  f=$(find gpurun_out/skab -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "$PAT" $f | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin): print('  ', r[0][:70].ljust(70), 'calls', r[1], 'avg_us', round(float(r[3])/1e3,2), 'min', round(float(r[5])/1e3,2), 'max', round(float(r[6])/1e3,2))"
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/skab.log
done; done
rm -rf gpurun_out/skab
