# per-phase cycle counts inside bwd_roles_kernel: a -DMP_ROLES_TIMING build, loaded (B = 32) and unloaded (B = 2) chip
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
OBJS=$(for f in *.hip; do [ $f != sa_bwd_fused.hip ] && echo $GRAFT_REPO_ROOT/maskplanner_amd/lib/obj/${f%.hip}.o; done)
mkdir -p /tmp/rt; hipcc -DMP_ROLES_TIMING ${EXTRA} -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c sa_bwd_fused.hip -o /tmp/rt/sa_bwd_fused.o 2>/dev/null
hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o /tmp/rt/lib.so /tmp/rt/sa_bwd_fused.o $OBJS
cd $GRAFT_REPO_ROOT
for b in 32 2; do MASKPLANNER_HIP_LIB=/tmp/rt/lib.so python tools/roles_timing.py $b 2>&1 | grep wave; done
