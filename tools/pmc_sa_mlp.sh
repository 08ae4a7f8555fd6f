# PMC passes over tools/prof_sa_mlp.py (counters only: no trace domains besides --kernel-trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU"; do
  i=$((i+1)); rm -rf gpurun_out/pmcm$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcm$i -- python3 tools/prof_sa_mlp.py > gpurun_out/pmcm$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2, 3):
    for f in glob.glob(f"gpurun_out/pmcm{i}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60], r["Grid_Size"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "gemm" not in k[0]:
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    act = m.get("GRBM_GUI_ACTIVE", 0)
    if not act:
        continue
    print(f"{k[0][:58]:58s} grid={k[1]:>9s} act={act/1e3:7.0f}k mfma_util={100*m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(act*1024):5.1f}% "
          f"wave_cyc/act/cu={m.get('SQ_WAVE_CYCLES',0)*4/act/256:5.2f} wait_any={100*m.get('SQ_WAIT_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1):5.1f}% "
          f"wait_inst={100*m.get('SQ_WAIT_INST_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1):5.1f}% valu_act={100*m.get('SQ_ACTIVE_INST_VALU',0)*4/(act*1024):5.1f}% "
          f"lds_act={100*m.get('SQ_ACTIVE_INST_LDS',0)*4/(act*1024):5.1f}% bank_conf={100*m.get('SQ_LDS_BANK_CONFLICT',0)/max(m.get('SQ_LDS_IDX_ACTIVE',1),1):5.1f}%")
PY
