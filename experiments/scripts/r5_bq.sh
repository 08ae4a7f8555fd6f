cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "ball_query or fps" 2>&1 | tail -5 > gpurun_out/r5q/tests.txt
python tools/bq_time.py > gpurun_out/r5q/bq_time.txt 2>&1
python -m pytest tests/test_gpu_modules.py tests/test_gpu_bf16.py -m gpu -x -q -k "plan_carried or config5 or msg" 2>&1 | tail -5 > gpurun_out/r5q/tests2.txt
cat gpurun_out/r5q/*.txt
