cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
rm -rf $O/pmcg_a $O/pmcg_b $O/pmcg_c
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmcg_a -- python3 tools/gemm_pmc_run.py > $O/pmcg_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmcg_b -- python3 tools/gemm_pmc_run.py > $O/pmcg_b.log 2>&1 || echo passb failed
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmcg_c -- python3 tools/gemm_pmc_run.py > $O/pmcg_c.log 2>&1 || echo passc failed
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/r3/pmcg_a", "gpurun_out/r3/pmcg_b", "gpurun_out/r3/pmcg_c"):
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs: print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "gemm16_kernel" not in k and "strip" not in k: continue
        name = k.replace("void ddpmir::(anonymous namespace)::", "").split("(")[0][:32] + " grid" + r["Grid_Size"]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in agg.items():
        print(name, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
