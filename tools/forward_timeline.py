"""Ordered per-dispatch listing of ONE U-Net forward from a rocprofv3 kernel trace (the middle forward of the run, i.e. one of
bench.py's timed steps: the dispatches between two consecutive nchw3_to_nhwc4 kernels), with the idle gap before each dispatch.
(Not the last forward: for the bottleneck-attention workloads that one belongs to the untimed GEMM-class pass, whose per-launch
HIP event records put ~5 us before and after every GEMM launch.)
usage: python tools/forward_timeline.py <rocprofv3 output dir> > profiles/rNN_forward_timeline.txt"""
import csv, glob, os, sys
d = sys.argv[1]
t = max(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "nchw3_to_nhwc4" in r["Kernel_Name"]]
a, b = starts[len(starts) // 2 - 1], starts[len(starts) // 2]
fwd = rows[a:b]
t0 = int(fwd[0]["Start_Timestamp"])
prev_end = t0
busy = gaps = 0.0
print(f"{'start us':>9s} {'gap us':>7s} {'dur us':>8s}  grid (workgroups)      kernel")
for r in fwd:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].replace("void ", "").replace("ddpmir::(anonymous namespace)::", "").replace("ddpmir::", "").split("(")[0]
    wg = [int(r[f"Grid_Size_{c}"]) // max(1, int(r[f"Workgroup_Size_{c}"])) for c in "XYZ"]
    gap = (s - prev_end) / 1e3
    busy += (e - s) / 1e3; gaps += max(0.0, gap)
    print(f"{(s - t0) / 1e3:9.1f} {gap:7.1f} {(e - s) / 1e3:8.1f}  {str(wg):22s} {k[:70]}")
    prev_end = max(prev_end, e)
print(f"{len(fwd)} dispatches, {busy / 1e3:.3f} ms inside kernels, {gaps / 1e3:.3f} ms of gaps between them, "
      f"{(prev_end - t0) / 1e6:.3f} ms wall")
