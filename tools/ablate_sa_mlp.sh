# per-shape kernel times of the SA MLP with parts of pos_gemm_kernel compiled out (maskplanner_amd/lib/ablate/lib_*.so:
# hipcc -DMP_ABLATE_MFMA | -DMP_ABLATE_EPI | -DMP_ABLATE_LOAD) -- results are wrong by construction, only times matter
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in BASE MFMA EPI LOAD; do
  if [ $v = BASE ]; then unset MASKPLANNER_HIP_LIB; else export MASKPLANNER_HIP_LIB=$GRAFT_REPO_ROOT/maskplanner_amd/lib/ablate/lib_$v.so; fi
  rm -rf gpurun_out/abl_$v
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abl_$v -- python3 tools/prof_sa_mlp.py > gpurun_out/abl_$v.log 2>&1
  echo "== $v"; python tools/ktrace.py gpurun_out/abl_$v pos_gemm | grep -v "grid=(8192\|grid=(16384"
done
