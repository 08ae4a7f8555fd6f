cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
python tools/dp_overhead.py 29641 30 > gpurun_out/r5d/dp_overhead.txt 2>&1
python -m pytest tests/test_gpu_dp.py tests/test_gpu_modules.py -m gpu -x -q -k "rccl or eight_ranks or dp_" 2>&1 | tail -5 > gpurun_out/r5d/tests_dp.txt
grep -v "^frame" gpurun_out/r5d/dp_overhead.txt | tail -5; cat gpurun_out/r5d/tests_dp.txt
