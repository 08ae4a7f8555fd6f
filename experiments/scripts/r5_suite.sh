cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r5s/tests.txt
cat gpurun_out/r5s/tests.txt
