# [r5] what do the dW atomic epilogues cost?  timing builds with them compiled out (results invalid): apply experiments/r5_no_dw_atomics_probe.patch first
cd $GRAFT_REPO_ROOT/maskplanner_amd/csrc
for v in 0 1; do
  d=/tmp/ap$v; mkdir -p $d
  for f in sa_bwd_fused sa_mlp; do
    hipcc -DMP_PROBE_NO_DW_ATOMICS=$v -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c $f.hip -o $d/$f.o 2>/dev/null &
  done
done
wait
OBJS=$(for f in *.hip; do [ $f != sa_mlp.hip ] && [ $f != sa_bwd_fused.hip ] && echo $GRAFT_REPO_ROOT/maskplanner_amd/lib/obj/${f%.hip}.o; done)
for v in 0 1; do hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o /tmp/ap$v/lib.so /tmp/ap$v/sa_mlp.o /tmp/ap$v/sa_bwd_fused.o $OBJS; done
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for v in 0 1; do
  echo -n "[no dW atomics=$v]: "; MASKPLANNER_HIP_LIB=/tmp/ap$v/lib.so python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys,re
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
sel={n: v for n,v in k.items() if re.search('bwd_roles|bwd_fused|dw_gemm', n)}
print(round(d['step_ms_median'],3), 'sum', round(sum(sel.values()),1), {n[:34]: round(v,1) for n,v in sorted(sel.items(), key=lambda kv: -kv[1])[:9]})"
done; done
