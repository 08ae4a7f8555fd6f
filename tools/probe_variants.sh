# VARS="-DA;-DB" bash tools/probe_variants.sh : tools/err_probe.py under compile-time variants of sa_mlp.hip (one box)
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
OBJS=$(for f in *.hip; do [ $f != sa_mlp.hip ] && echo $GRAFT_REPO_ROOT/maskplanner_amd/lib/obj/${f%.hip}.o; done)
IFS=';' read -ra VV <<< "${VARS}"
n=0
for v in "${VV[@]}"; do
  d=/tmp/pv$n; mkdir -p $d
  hipcc $v -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c sa_mlp.hip -o $d/sa_mlp.o 2>/dev/null &
  n=$((n+1))
done
wait
n=0; for v in "${VV[@]}"; do d=/tmp/pv$n; hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $d/lib.so $d/sa_mlp.o $OBJS; n=$((n+1)); done
cd $GRAFT_REPO_ROOT
n=0; for v in "${VV[@]}"; do echo "[$v]"; MASKPLANNER_HIP_LIB=/tmp/pv$n/lib.so python tools/err_probe.py $REPS 2>&1 | tail -1; n=$((n+1)); done
