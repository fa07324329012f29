cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc32a -- python3 tools/attn_time32.py > gpurun_out/pmc32a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc32b -- python3 tools/attn_time32.py > gpurun_out/pmc32b.log 2>&1 || echo passb failed
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc32a", "gpurun_out/pmc32b"):
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs: print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "flash_bf16" not in k: continue
        name = k.split("flash_bf16")[1].split("(")[0][:24] + " grid" + r["Grid_Size"]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in agg.items():
        print(name, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
