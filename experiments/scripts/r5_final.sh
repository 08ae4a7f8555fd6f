# [r5] end-of-round evidence in one call: the GPU suite, the rocprof / PMC summaries, the bench lines of every config
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5s
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r5s/tests.txt
cat gpurun_out/r5s/tests.txt | cut -c1-400
bash tools/refresh_profiles.sh r05 > gpurun_out/r05_refresh.log 2>&1
bash tools/refresh_lines.sh r05 > gpurun_out/r05_refresh_lines.log 2>&1
tail -c 400 gpurun_out/r05_bench_full.log
