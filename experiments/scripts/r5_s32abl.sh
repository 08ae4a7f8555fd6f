# [r5] where a chunk of bwd_stream32_kernel goes: timing-only builds with phases removed (results invalid), kernel time from bench.py
cd $GRAFT_REPO_ROOT
cd maskplanner_amd/csrc
OBJS=$(for f in *.hip; do [ $f != sa_stream16.hip ] && echo $GRAFT_REPO_ROOT/maskplanner_amd/lib/obj/${f%.hip}.o; done)
n=0
ABL="-DMP_S32_NO_DW -DMP_S32_NO_DX -DMP_S32_NO_STAGE"
B="$ABL -DMP_S32_NO_ST -DMP_S32_INTERLEAVE=0 -DMP_S32_ROT=13"
for v in "$B" "$B -DMP_GLDS_NT" "$B -DMP_S32_NO_P" "$B -DMP_S32_NO_P -DMP_GLDS_NT" "$ABL -DMP_S32_INTERLEAVE=0 -DMP_S32_ROT=13 -DMP_GLDS_NT" "-DMP_S32_INTERLEAVE=0 -DMP_S32_ROT=13 -DMP_GLDS_NT"; do
  d=/tmp/sv$n; mkdir -p $d
  hipcc $v -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c sa_stream16.hip -o $d/s.o 2>/dev/null &
  n=$((n+1))
done
wait
cd $GRAFT_REPO_ROOT
for i in 0 1 2 3 4 5; do
  hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o /tmp/sv$i/lib.so /tmp/sv$i/s.o $OBJS
  echo -n "[variant $i]: "; MASKPLANNER_HIP_LIB=/tmp/sv$i/lib.so timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys,re
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d['kernels_us_per_step']
print({n[:40]: round(v,1) for n,v in k.items() if 'stream32' in n})"
done
