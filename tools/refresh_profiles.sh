# everything under profiles/<tag>_* from one GPU session: kernel stats + PMC passes of the default bench, the overlap trace, the SQ counter
# table of the hot kernels, the full default bench line and the lines of the other BASELINE configs.   bash tools/refresh_profiles.sh r02
TAG=${1:-r04}
cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh $TAG > gpurun_out/profile_round.log 2>&1
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/${TAG}_overlap
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_overlap -- python3 bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-side-legs > gpurun_out/${TAG}_overlap.log 2>&1
find gpurun_out/${TAG}_overlap -type f ! -name "*kernel_trace.csv" -delete
python3 tools/make_overlap_summary.py $TAG gpurun_out/${TAG}_overlap > gpurun_out/${TAG}_overlap_summary.log 2>&1
bash tools/pmc_hot.sh > gpurun_out/${TAG}_pmc_hot.txt 2>&1
python3 bench.py > gpurun_out/${TAG}_bench_full.log 2>&1
python3 bench.py --category windows --no-cpu-baseline --no-side-legs > gpurun_out/${TAG}_windows.log 2>&1
python3 bench.py --category shelves --no-cpu-baseline --no-side-legs > gpurun_out/${TAG}_shelves.log 2>&1
python3 bench.py --category containers --points 10240 --encoder msg --dtype bf16 --steps 40 --no-cpu-baseline --no-side-legs > gpurun_out/${TAG}_c5_bf16.log 2>&1
python3 bench.py --category containers --points 10240 --encoder msg --dtype f32 --steps 40 --no-cpu-baseline --no-side-legs > gpurun_out/${TAG}_c5_f32.log 2>&1
MP_SA_SPLIT=0 MP_KNN_SCREEN=0 python3 bench.py --no-cpu-baseline --no-side-legs > gpurun_out/${TAG}_fp32mfma.log 2>&1
tail -c 300 gpurun_out/${TAG}_bench_full.log
