# per-kernel times of the set-abstraction MLPs with and without MP_SA_SPLIT (rocprofv3 --stats), on the GPU box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 1; do
  export MP_SA_SPLIT=$v
  rm -rf gpurun_out/split_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/split_$v -- python3 tools/prof_sa_mlp.py > /dev/null 2>&1
  echo "MP_SA_SPLIT=$v"
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/split_$v/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    n = r["Name"].replace("void (anonymous namespace)::", "").split("(")[0]
    if any(k in n for k in ("fwd_chunk", "bwd_fused", "bwd_first", "pos_gemm", "dw_gemm", "dw_ci4", "rc_stats")):
        print(f'{float(r["AverageNs"]) / 1e3:8.1f} us x{r["Calls"]:>3s}  {n}')
PY
done
