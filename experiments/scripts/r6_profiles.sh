# round 6: everything under profiles/r06_* from one GPU session (then `python tools/collect_profiles.py r06` in the build container)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
python -m pytest tests/ -x -q -m gpu > gpurun_out/r6/full_gpu.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r6/full_gpu.txt
bash tools/refresh_profiles.sh r06 > gpurun_out/r06_refresh.log 2>&1
bash tools/profile_configs.sh r06 > gpurun_out/r06_profile_configs.log 2>&1
PMC_ARGS="--encoder msg --category containers --points 10240 --dtype bf16" bash tools/pmc_hot.sh > gpurun_out/r06_pmc_hot_c5.txt 2>&1
# the driver's own command line, twice
for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_driver_line_$i.log 2>&1; done
tail -c 600 gpurun_out/r06_refresh.log
du -sh gpurun_out
