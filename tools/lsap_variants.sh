# VARS="-DA;-DB" bash tools/lsap_variants.sh : lsap.hip compile-time variants: the LAP tests and tools/bench_lsap.py under each (one box)
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
OBJS=$(for f in *.hip; do [ $f != lsap.hip ] && echo $GRAFT_REPO_ROOT/maskplanner_amd/lib/obj/${f%.hip}.o; done)
IFS=';' read -ra VV <<< "${VARS}"
n=0
for v in "${VV[@]}"; do
  d=/tmp/lv$n; mkdir -p $d
  hipcc $v -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c lsap.hip -o $d/l.o 2>&1 | grep -E "error" &
  n=$((n+1))
done
wait
n=0; for v in "${VV[@]}"; do d=/tmp/lv$n; hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $d/lib.so $d/l.o $OBJS; n=$((n+1)); done
cd $GRAFT_REPO_ROOT
n=0; for v in "${VV[@]}"; do echo "[$v]"; MASKPLANNER_HIP_LIB=/tmp/lv$n/lib.so python -m pytest tests/test_gpu_ops.py -q -m gpu -k lsap 2>&1 | tail -1
  for r in 1 2; do MASKPLANNER_HIP_LIB=/tmp/lv$n/lib.so python tools/bench_lsap.py 2>&1 | tail -1 | cut -c1-50; done; n=$((n+1)); done
