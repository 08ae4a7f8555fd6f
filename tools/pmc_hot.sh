#!/bin/bash
# SQ counter passes over the default bench for the hot grouped-MLP kernels (counters only; each group in its own run)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAIT_INST_LDS"; do
  i=$((i+1)); rm -rf gpurun_out/pmch$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmch$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-legs $PMC_ARGS > gpurun_out/pmch$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2, 3, 4):
    for f in glob.glob(f"gpurun_out/pmch{i}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':48s} {'act':>8s} {'mfma%':>6s} {'waves/simd':>10s} {'wait_any%':>9s} {'wait_inst%':>10s} {'valu%':>6s} {'lds%':>6s} {'vmem%':>6s} {'bankc%':>7s} {'waitlds%':>8s} {'n_valu':>9s} {'n_lds':>8s} {'n_mfma':>8s}")
for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    if not any(t in k for t in ("fused", "roles", "chunk", "bwd_first", "pos_gemm", "dw_gemm", "stream16", "bwd_pair")):
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    act = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if not act:
        continue
    wc = m.get("SQ_WAVE_CYCLES", 0) * 4          # quad-cycles -> cycles, summed over waves
    print(f"{k:48s} {act:8.0f} {100*m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*act):6.1f} {wc/(1024*act):10.2f} "
          f"{100*m.get('SQ_WAIT_ANY',0)*4/max(wc,1):9.1f} {100*m.get('SQ_WAIT_INST_ANY',0)*4/max(wc,1):10.1f} "
          f"{100*m.get('SQ_ACTIVE_INST_VALU',0)*4/(1024*act):6.1f} {100*m.get('SQ_ACTIVE_INST_LDS',0)*4/(1024*act):6.1f} "
          f"{100*m.get('SQ_ACTIVE_INST_VMEM',0)*4/(1024*act):6.1f} {100*m.get('SQ_LDS_BANK_CONFLICT',0)/max(m.get('SQ_LDS_IDX_ACTIVE',1),1):7.1f} "
          f"{100*m.get('SQ_WAIT_INST_LDS',0)*4/max(wc,1):8.1f} {m.get('SQ_INSTS_VALU',0):9.3g} {m.get('SQ_INSTS_LDS',0):8.3g} {m.get('SQ_INSTS_MFMA',0):8.3g}")
PY
find gpurun_out/pmch* -type f ! -name '*.csv' -delete 2>/dev/null; find gpurun_out/pmch* -name '*kernel_trace.csv' -delete 2>/dev/null
