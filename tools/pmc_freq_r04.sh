# SQ counters of the fused frequency-gate kernel (one pass) under tools/micro/freq_phase: parked / issue-stalled / issuing shares of
# its wave cycles and the VALU's share of the issuing ones.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4pmcf; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/p -- ./tools/micro/freq_phase 8 1 > $O/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/r4pmcf/p/*/*counter_collection.csv")
agg = collections.defaultdict(dict)
for r in csv.DictReader(open(fs[0])):
    if "freq_gate_fused" not in r["Kernel_Name"]: continue
    agg[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
g = list(agg.values())
avg = {c: sum(x[c] for x in g) / len(g) for c in g[0]}
wc = avg["SQ_WAVE_CYCLES"]
print(f"freq_gate_fused_k<8,64,false>, 512 x 512 x 8, bf16 in / out, {len(g)} launches averaged")
print({c: round(v) for c, v in avg.items()})
print(f"of the wave cycles: parked at s_waitcnt / barrier {avg['SQ_WAIT_ANY'] / wc:.3f}, issue-stalled {avg['SQ_WAIT_INST_ANY'] / wc:.3f}, issuing {avg['SQ_ACTIVE_INST_ANY'] / wc:.3f}"
      f" (VALU {avg['SQ_ACTIVE_INST_VALU'] / wc:.3f}, LDS {avg['SQ_ACTIVE_INST_LDS'] / wc:.3f})")
dur = avg["GRBM_GUI_ACTIVE"] / 8
print(f"duration {dur:,.0f} cycles; vector instructions per wave {avg['SQ_INSTS_VALU'] / (16384 * 4):.0f}; SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = {avg['SQ_ACTIVE_INST_VALU'] / avg['SQ_INSTS_VALU']:.2f}"
      f" (the SQ counts in units of 4 cycles: one unit per wave64 vector instruction) -> VALU busy {4 * avg['SQ_ACTIVE_INST_VALU'] / 1024 / dur:.2f} of the launch on every SIMD")
PY
