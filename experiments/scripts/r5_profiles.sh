cd "$GRAFT_REPO_ROOT"
bash tools/refresh_profiles.sh r05 > gpurun_out/r05_refresh.log 2>&1
bash tools/profile_configs.sh r05 > gpurun_out/r05_profile_configs.log 2>&1
PMC_ARGS="--encoder msg --category containers --points 10240 --dtype bf16" bash tools/pmc_hot.sh > gpurun_out/r05_pmc_hot_c5.txt 2>&1
tail -c 600 gpurun_out/r05_refresh.log
