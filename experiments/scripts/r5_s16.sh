cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r5c/tests_bf16.txt
cat gpurun_out/r5c/tests_bf16.txt
A="--encoder msg --category containers --points 10240 --dtype bf16 --no-cpu-baseline --no-side-legs --steps 40 --warmup 5"
for i in 1 2; do for v in 1 2 0; do
  MP_S16=$v timeout 300 python3 bench.py $A 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
print('MP_S16=$v', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median', 0),3), {n.replace('_kernel','').replace('stream16','s16')[:30]: round(x,1) for n,x in k.items() if ('bwd_' in n or 'fwd_' in n) and x > 30})
"
done; done > gpurun_out/r5c/ab.txt 2>&1
cat gpurun_out/r5c/ab.txt
