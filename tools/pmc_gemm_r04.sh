# round 4: SQ counters (one pass) of the dominant 3x3 conv shapes of configs[4] with the current kernels: the strip kernel with the
# register-direct epilogue (bf16 output only = conv2 of a block, and fp32 + bf16), the 128-column kernel, the 256 x 256 eight-wave tile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4pmc; mkdir -p $O
rm -rf $O/pmcg_a
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmcg_a -- python3 tools/gemm_pmc_run.py r04 > $O/pmcg_a.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/r4pmc/pmcg_a/*/*counter_collection.csv")
agg = collections.OrderedDict()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"]
    if not any(t in k for t in ("gemm16_kernel", "strip", "gemm256_kernel")): continue
    name = k.replace("void ddpmir::(anonymous namespace)::", "").split("(")[0][:34] + " grid" + r["Grid_Size"] + " #" + str(int(r["Dispatch_Id"]) // 4 if False else "")
    key = (name, r["Dispatch_Id"])
    agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
# average per (kernel, grid) in launch order groups of equal names
order, groups = [], collections.defaultdict(list)
for (name, did), cs in agg.items():
    if name not in groups: order.append(name)
    groups[name].append(cs)
for name in order:
    g = groups[name]
    keys = g[0].keys()
    avg = {c: sum(x[c] for x in g) / len(g) for c in keys}
    wait = avg.get("SQ_WAIT_ANY", 0) / max(1.0, avg.get("SQ_WAVE_CYCLES", 1))
    dur = avg.get("GRBM_GUI_ACTIVE", 0) / 8
    mfma = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, dur * 1024)
    print(f"{name} x{len(g)}: duration {dur:,.0f} cycles, SQ_WAIT_ANY / SQ_WAVE_CYCLES = {wait:.3f}, matrix pipe busy {mfma:.3f}, " + str({c: round(v) for c, v in avg.items()}))
PY
