# phase times inside mask_match_kernel: a -DMP_MM_TIMING build of the library next to the shipped one, on one box
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
d=/tmp/mmt; mkdir -p $d
hipcc -DMP_MM_TIMING $MM_DEFS -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -I../../include -c mask_match.hip -o $d/mask_match.o || exit 1
hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $d/lib.so $d/mask_match.o $(ls ../lib/obj/*.o | grep -v mask_match.o)
cd $GRAFT_REPO_ROOT
MASKPLANNER_HIP_LIB=$d/lib.so python tools/mask_match_time.py --phases 2>&1 | tail -8
