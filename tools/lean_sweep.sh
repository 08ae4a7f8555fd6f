# per-kernel times of the lean kernels for every library variant built by tools/lean_variants.sh (GPU box)
export MP_LEAN_LAST=1
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  name=${v%%:*}; ppb=${v##*:}
  lib=$GRAFT_REPO_ROOT/maskplanner_amd/lib/var_$name.so
  [ "$name" = base ] && lib=$GRAFT_REPO_ROOT/maskplanner_amd/lib/libmaskplanner_hip.so
  rm -rf /tmp/lp_$name
  MASKPLANNER_HIP_LIB=$lib MP_LEAN_PPB=$ppb rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp_$name -- python3 $GRAFT_REPO_ROOT/tools/lean_time.py > /dev/null 2>&1
  f=$(ls /tmp/lp_$name/*/*kernel_stats.csv | head -1)
  python3 - "$f" "$name:$ppb" <<EOF
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print(sys.argv[2], " ".join(f"{r['Name'].split('(')[0].split('::')[-1]}={float(r['AverageNs'])/1e3:.0f}" for r in rows if "lean_d" in r["Name"]))
EOF
done
