# the bench lines kept under profiles/<tag>_*.json, re-run once profiles/<tag>_traffic.json exists (so that `roofline.traffic` of every line is
# the measured figure of its own config):  bash tools/refresh_lines.sh r03 ; python tools/collect_profiles.py r03 --lines-only
TAG=${1:-r03}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python3 bench.py > gpurun_out/${TAG}_bench_full.log 2>&1
python3 bench.py --category windows --no-cpu-baseline --no-side-legs > gpurun_out/${TAG}_windows.log 2>&1
python3 bench.py --category shelves --no-cpu-baseline --no-side-legs > gpurun_out/${TAG}_shelves.log 2>&1
python3 bench.py --category containers --points 10240 --encoder msg --dtype bf16 --steps 40 --no-cpu-baseline --no-side-legs > gpurun_out/${TAG}_c5_bf16.log 2>&1
python3 bench.py --category containers --points 10240 --encoder msg --dtype f32 --steps 40 --no-cpu-baseline --no-side-legs > gpurun_out/${TAG}_c5_f32.log 2>&1
MP_SA_SPLIT=0 MP_KNN_SCREEN=0 python3 bench.py --no-cpu-baseline --no-side-legs > gpurun_out/${TAG}_fp32mfma.log 2>&1
python3 tools/dropin_phases.py 2>&1 | grep -v Warn > gpurun_out/${TAG}_dropin_phases.log
tail -c 300 gpurun_out/${TAG}_bench_full.log
