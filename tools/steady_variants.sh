# VARS="-DA;-DB" bash tools/steady_variants.sh : the run-to-run steadiness test under compile-time variants of sa_mlp.hip (one box)
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
OBJS=$(for f in *.hip; do [ $f != sa_mlp.hip ] && echo $GRAFT_REPO_ROOT/maskplanner_amd/lib/obj/${f%.hip}.o; done)
IFS=';' read -ra VV <<< "${VARS}"
n=0
for v in "${VV[@]}"; do
  d=/tmp/tv$n; mkdir -p $d
  hipcc $v -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c sa_mlp.hip -o $d/sa_mlp.o 2>/dev/null &
  n=$((n+1))
done
wait
n=0; for v in "${VV[@]}"; do d=/tmp/tv$n; hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $d/lib.so $d/sa_mlp.o $OBJS; n=$((n+1)); done
cd $GRAFT_REPO_ROOT
n=0; for v in "${VV[@]}"; do echo "[$v]"; for r in 1 2; do MASKPLANNER_HIP_LIB=/tmp/tv$n/lib.so python -m pytest tests/test_gpu_bf16.py -q -k "steady" 2>&1 | tail -1; done; n=$((n+1)); done
