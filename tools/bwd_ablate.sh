# where an iteration of the fused backward kernels goes: timing builds with parts compiled out (MP_BF_ABL bit mask; MP_DESYNC=0 so that the
# plain loop is what is measured), SA2 / SA1 shapes at B = 32 and B = 2 (tools/chunk_latency_probe.py)
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
OBJS=$(for f in *.hip; do [ $f != sa_bwd_fused.hip ] && echo $GRAFT_REPO_ROOT/maskplanner_amd/lib/obj/${f%.hip}.o; done)
VARS="${VARS:-0 1 2 4 8 16 32 3 59}"
n=0
for v in $VARS; do
  d=/tmp/abl$v; mkdir -p $d
  hipcc -DMP_BF_ABL=$v -DMP_DESYNC=0 ${EXTRA} -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c sa_bwd_fused.hip -o $d/sa_bwd_fused.o 2>/dev/null &
  n=$((n+1)); [ $((n % 6)) = 0 ] && wait
done
wait
cd $GRAFT_REPO_ROOT
for v in $VARS; do d=/tmp/abl$v; hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $d/lib.so $d/sa_bwd_fused.o $OBJS
  echo "== MP_BF_ABL=$v"; MASKPLANNER_HIP_LIB=$d/lib.so python tools/chunk_latency_probe.py 32 2 2>&1 | grep "^S=" | sed -e 's/fwd_chunk[^ ]* [^ ]* [0-9.]*//g' | cut -c1-230
done
