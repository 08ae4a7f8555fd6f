cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5k
for q in "" 8 16; do
  if [ -z "$q" ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for i in 1 2; do
  python tools/dp_overhead.py $((29650 + i)) 30 2>/dev/null | grep "^{" | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('HWQ', os.environ.get('GPU_MAX_HW_QUEUES'), 'n1', round(d['n1_path_ms'],3), 'dp', round(d['dp_path_one_rank_ms'],3), d['rounds'])"
  done
done > gpurun_out/r5k/hwq_dp.txt 2>&1
cat gpurun_out/r5k/hwq_dp.txt
