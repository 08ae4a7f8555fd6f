cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5c
SRC=sa_stream16.hip VARS="-DMP_S16_RING=3 -DMP_S16_FRING=3;-DMP_S16_RING=2 -DMP_S16_FRING=2;-DMP_S16_RING=3 -DMP_S16_FRING=4" FILTER="s16|stream16" TESTS="tests/test_gpu_bf16.py" BENCH_ARGS="--encoder msg --category containers --points 10240 --dtype bf16" bash tools/sa_variants.sh > gpurun_out/r5c/variants.txt 2>&1
cat gpurun_out/r5c/variants.txt
