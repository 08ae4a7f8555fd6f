# A/B of the fused backward kernels on one box, alternating: the row-swizzled K-packed image (MP_KSWZ) and the two halves of the 512-thread
# workgroups walking the matrix / staging phases out of step (MP_DESYNC); correctness of the full variant first
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
OBJS=$(for f in *.hip; do [ $f != sa_mlp.hip ] && echo ../lib/obj/${f%.hip}.o; done)
for v in 00 01 11; do
  d=/tmp/ks$v; mkdir -p $d
  hipcc -DMP_KSWZ=${v:0:1} -DMP_DESYNC=${v:1:1} -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c sa_mlp.hip -o $d/sa_mlp.o 2>/dev/null &
done
wait
for v in 00 01 11; do d=/tmp/ks$v; hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $d/lib.so $d/sa_mlp.o $OBJS; done
cd $GRAFT_REPO_ROOT
for v in 01 11; do MASKPLANNER_HIP_LIB=/tmp/ks$v/lib.so python -m pytest tests/test_gpu_split.py tests/test_gpu_bf16.py tests/test_gpu_modules.py -q -x 2>&1 | tail -3; done
for i in 1 2 3; do for v in 00 01 11; do
  echo -n "kswz,desync=$v: "; MASKPLANNER_HIP_LIB=/tmp/ks$v/lib.so python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
print(round(d['ms_per_step'],3), round(d.get('step_ms_median',0),3), {n[:28]: round(v,1) for n,v in k.items() if 'bwd_f' in n or 'fwd_chunk_kernel<128, 256' in n})"
done; done
