#!/bin/bash
# Step-boundary bubble of the replayed step: host-side probe + kernel traces of the default step and of the variants without
# the eager launches between the graphs.  (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bgap; rm -rf $O; mkdir -p $O
python3 tools/boundary_probe.py 60 > $O/probe_default.txt 2>&1
MASKPLANNER_OVERLAP_SAMPLING=0 python3 tools/boundary_probe.py 60 > $O/probe_inline_sampling.txt 2>&1
MASKPLANNER_OVERLAP_SAMPLING=0 MASKPLANNER_SPLIT_ADAM=0 python3 tools/boundary_probe.py 60 > $O/probe_one_graph.txt 2>&1
for v in default inline one; do
  case $v in
    default) E="";;
    inline) E="MASKPLANNER_OVERLAP_SAMPLING=0";;
    one) E="MASKPLANNER_OVERLAP_SAMPLING=0 MASKPLANNER_SPLIT_ADAM=0";;
  esac
  for kv in $E; do export $kv; done
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_$v -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-legs > $O/tr_$v.log 2>&1
  unset MASKPLANNER_OVERLAP_SAMPLING MASKPLANNER_SPLIT_ADAM
  python3 tools/step_sequence.py $O/tr_$v 3 > $O/seq_$v.txt 2>&1
  find $O/tr_$v -type f ! -name '*kernel_trace.csv' -delete
done
python3 bench.py --no-cpu-baseline --no-side-legs > $O/bench.json 2> $O/bench.err
tail -n 12 $O/probe_*.txt; grep -h "span" $O/seq_*.txt; awk '$5>3' $O/seq_default.txt | head; cut -c1-400 $O/bench.json
