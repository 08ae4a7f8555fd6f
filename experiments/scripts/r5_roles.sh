cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5r
VARS="-DMP_ROLES_BEARLY=0;-DMP_ROLES_BEARLY=1;-DMP_ROLES_MFMA_ORDER=1;-DMP_ROLES_PD2=1" FILTER="bwd_roles" TESTS="tests/test_gpu_split.py tests/test_gpu_bnsites.py" bash tools/sa_variants.sh > gpurun_out/r5r/variants.txt 2>&1
EXTRA="-DMP_ROLES_PD2=1" bash tools/roles_timing.sh > gpurun_out/r5r/roles_timing_pd2.txt 2>&1
EXTRA= bash tools/roles_timing.sh > gpurun_out/r5r/roles_timing.txt 2>&1
python tools/stream_gap.py > gpurun_out/r5r/stream_gap.txt 2>&1
python -m pytest tests/test_gpu_modules.py -m gpu -x -q -k "stream or plan_carried" 2>&1 | tail -3 > gpurun_out/r5r/tests.txt
cat gpurun_out/r5r/tests.txt gpurun_out/r5r/variants.txt gpurun_out/r5r/roles_timing_pd2.txt gpurun_out/r5r/roles_timing.txt; tail -2 gpurun_out/r5r/stream_gap.txt
