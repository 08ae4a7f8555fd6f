cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5t
python -m pytest tests/test_gpu_modules.py -m gpu -x -q -k "stream or plan_carried or collat" 2>&1 | tail -3 > gpurun_out/r5t/tests.txt
python tools/stream_gap.py 2>/dev/null | tail -1 > gpurun_out/r5t/pieces.txt
cat gpurun_out/r5t/tests.txt gpurun_out/r5t/pieces.txt
