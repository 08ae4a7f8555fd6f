for i in 1 2 3; do
for v in cur prev; do
  if [ $v = prev ]; then export MASKPLANNER_HIP_LIB=$GRAFT_REPO_ROOT/maskplanner_amd/lib/ablate/lib_prev.so; else unset MASKPLANNER_HIP_LIB; fi
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); print('$v', round(d['ms_per_step'],3))"
done; done
