# [r5] B = 64 / B = 1 / dropout-order tests, then the default bench with the new side legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
python -m pytest tests/test_gpu_heads.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r5b/tests_heads.txt
python -m pytest tests/test_gpu_modules.py -m gpu -x -q -k "reference_batch_size_64 or batch_of_one or plan_carried" 2>&1 | tail -8 > gpurun_out/r5b/tests_modules.txt
python bench.py --steps 40 --warmup 5 > gpurun_out/r5b/bench.json 2> gpurun_out/r5b/bench.err
tail -c 1500 gpurun_out/r5b/bench.err
