"""Per-kernel HBM-side traffic of the bench workload from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).
usage: pmc_summary.py <fetch_dir> <write_dir> <n_forwards> [--json FILE --key WORKLOAD_KEY --command "bench command" --class gemm]
(--class gemm: the roofline "kernel" is the GEMM class -- every gemm16_kernel / gemm256_kernel / conv3x3_strip_kernel / splitk_reduce_k launch of
one forward -- as bench.py reports for the workloads with attention at the bottleneck only)
FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B); both counters are in KiB.  Durations are the dispatch
timestamps of the same (profiled) runs.  With --json the bytes per launch of the roofline kernel (the level-0 attention
dispatches: the flash kernel launches with the largest grid) are merged into FILE under WORKLOAD_KEY -- that is what
bench.py reports as roofline.traffic, with this file as its stated source."""
import collections
import csv
import glob
import sys


def load(d, counter):
    import os
    f = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)   # newest run in the directory
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("ddpmir::(anonymous namespace)::", "").split("(")[0]
        a = agg[k]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return agg


def roofline_kernel_bytes(d, counter):
    """mean counter value (KiB) over the level-0 attention dispatches: flash kernel launches with the largest grid"""
    import os
    f = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "flash_bf16" in r["Kernel_Name"]]
    if not rows:
        return None, None, 0
    if any("flash_bf16_nm" in r["Kernel_Name"] for r in rows):   # the no-max kernel does the work of the bounded pairs
        rows = [r for r in rows if "flash_bf16_nm" in r["Kernel_Name"]]
    gmax = max(int(r["Grid_Size"]) for r in rows)
    sel = [r for r in rows if int(r["Grid_Size"]) == gmax]
    name = sel[0]["Kernel_Name"].replace("void ", "").replace("ddpmir::(anonymous namespace)::", "").split("(")[0]
    return sum(float(r["Counter_Value"]) for r in sel) / len(sel), name, len(sel)


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
n = int(sys.argv[3])
if n <= 0:   # one image-layout kernel per U-Net forward
    n = max(1, next((v[0] for k, v in fe.items() if "nchw3_to_nhwc4" in k), 1))
print(f"{'kernel':58s} {'calls/fwd':>9s} {'ms/fwd':>8s} {'read GB':>8s} {'write GB':>8s} {'GB/s':>7s}   (per forward; read = FETCH_SIZE x2)")
tot = [0.0, 0.0, 0.0]
setup = None
for k in sorted(fe, key=lambda k: -fe[k][2]):
    c, f_kib, ms = fe[k]
    w_kib = wr.get(k, [0, 0.0, 0.0])[1]
    if "fillBufferAligned" in k:   # hipMemset of the workspace arena and of the weight blob: once per context, not per forward
        setup = (c, w_kib * 1024 / 1e9, ms)
        continue
    rd, wrb = 2 * f_kib * 1024 / 1e9 / n, w_kib * 1024 / 1e9 / n
    ms /= n
    tot[0] += rd; tot[1] += wrb; tot[2] += ms
    if ms < 0.05:
        continue
    print(f"{k[:58]:58s} {c / n:9.1f} {ms:8.3f} {rd:8.3f} {wrb:8.3f} {(rd + wrb) / (ms / 1e3):7.0f}")
print(f"{'all kernels':58s} {'':9s} {tot[2]:8.3f} {tot[0]:8.3f} {tot[1]:8.3f} {(tot[0] + tot[1]) / (tot[2] / 1e3):7.0f}")
if setup:
    print(f"(one-time setup, not in the table: {setup[0]} hipMemset launches clearing the workspace arena / weight blob, {setup[1]:.2f} GB, {setup[2]:.2f} ms)")

if "--json" in sys.argv:
    import json, os, subprocess
    out = sys.argv[sys.argv.index("--json") + 1]
    key = sys.argv[sys.argv.index("--key") + 1]
    cmd = sys.argv[sys.argv.index("--command") + 1] if "--command" in sys.argv else None
    if "--class" in sys.argv and sys.argv[sys.argv.index("--class") + 1] == "gemm":
        sel = [k for k in fe if ("gemm16_kernel" in k or "gemm256_kernel" in k or "conv3x3_strip" in k or "splitk_reduce" in k or "conv_tile_kernel" in k or "head_fused_k" in k)]
        f_kib = sum(fe[k][1] for k in sel) / n
        w_kib = sum(wr.get(k, [0, 0.0, 0.0])[1] for k in sel) / n
        name, nl = "GEMM class: " + ", ".join(sorted(set(k.split("<")[0] for k in sel))), int(sum(fe[k][0] for k in sel) / n)
    else:
        f_kib, name, nl = roofline_kernel_bytes(sys.argv[1], "FETCH_SIZE")
        w_kib, _, _ = roofline_kernel_bytes(sys.argv[2], "WRITE_SIZE")
    doc = json.load(open(out)) if os.path.exists(out) else {"workloads": {}}
    try:   # the source hash the measured library was built from (ddpmir_version()): bench.py marks the figure stale when it differs
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from ddpm_image_restoration_amd import build as _b
        src_hash = _b.source_hash()
    except Exception:
        src_hash = None
    try:
        doc["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or doc.get("commit")
    except Exception:
        pass
    if f_kib is not None and w_kib is not None:
        doc["workloads"][key] = {"roofline_kernel": name, "launches_averaged": nl,
                                 "fetch_bytes_per_launch": 2 * f_kib * 1024, "write_bytes_per_launch": w_kib * 1024,
                                 "roofline_kernel_bytes_per_launch": 2 * f_kib * 1024 + w_kib * 1024,
                                 "all_kernels_read_GB_per_forward": tot[0], "all_kernels_write_GB_per_forward": tot[1],
                                 "command": cmd, "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes",
                                 "src_hash": src_hash}
        json.dump(doc, open(out, "w"), indent=1)
        print(f"merged {key} into {out}: {doc['workloads'][key]['roofline_kernel_bytes_per_launch'] / 1e9:.3f} GB per level-0 launch")
