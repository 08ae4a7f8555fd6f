# builds libmaskplanner_hip variants with other lean_dx settings (CPU container; the .so files travel with gpurun):
#   bash tools/lean_variants.sh "R4W4 -DMP_LEAN_RMAX=4 -DMP_LEAN_DX_WAVES=4" "R8W2 -DMP_LEAN_RMAX=8 -DMP_LEAN_DX_WAVES=2" ...
cd "$(dirname "$0")/../maskplanner_amd/csrc"
for spec in "$@"; do
  set -- $spec; name=$1; shift
  hipcc $* -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage \
     -c sa_lean.hip -o ../lib/obj/sa_lean_$name.o 2>&1 | grep -E "error|Spill: [1-9]|VGPRs: " | grep -B1 "Spill: [1-9]"
  objs=$(ls ../lib/obj/*.o | grep -v "sa_lean")
  hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o ../lib/var_$name.so $objs ../lib/obj/sa_lean_$name.o
  echo built var_$name.so
done
