# timing builds of the screened nearest-neighbour kernel with parts removed (results are wrong by construction):
#   1 no exact evaluation, 2 no candidate recording either, 3 no MFMA / block loop (tiles are still staged)
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
for v in 1 2 3; do
  mkdir -p /tmp/abl$v
  hipcc -DKS_ABL=$v -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c knn.hip -o /tmp/abl$v/knn.o
  hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o /tmp/abl$v/lib.so /tmp/abl$v/knn.o $(ls ../lib/obj/*.o | grep -v knn.o)
  echo "KS_ABL=$v"; (cd $GRAFT_REPO_ROOT && MASKPLANNER_HIP_LIB=/tmp/abl$v/lib.so timeout 200 python tools/knn_check.py 2>&1 | tail -4 | cut -c1-110)
done
