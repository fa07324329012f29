import csv, sys, collections, glob
d = sys.argv[1]
import os
t = max(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime)   # newest run in the directory
rows = list(csv.DictReader(open(t)))
n_fwd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if n_fwd <= 0:   # one image-layout kernel per U-Net forward
    n_fwd = max(1, sum(1 for r in rows if "nchw3_to_nhwc4" in r["Kernel_Name"]))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r["Kernel_Name"].replace("void ", "").replace("ddpmir::(anonymous namespace)::", "").split("(")[0]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[k][0] += 1; agg[k][1] += us
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot/1e3:.2f} ms over {n_fwd} forwards = {tot/1e3/n_fwd:.2f} ms/forward")
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{k[:60]:60s} calls/fwd={c/n_fwd:6.1f} ms/fwd={us/1e3/n_fwd:8.3f}")
# the roofline kernel: level-0 attention launches only (the largest grid of flash_bf16<16,...>); bench.py's
# roofline.avg_launch_ms is the HIP-event average of exactly these dispatches
fl = [r for r in rows if "flash_bf16" in r["Kernel_Name"]]
if any("flash_bf16_nm" in r["Kernel_Name"] for r in fl):   # bounded scores run in the no-max kernel; the general kernel's
    fl = [r for r in fl if "flash_bf16_nm" in r["Kernel_Name"]]   # workgroups of those pairs return at once
if fl:
    gmax = max(int(r["Grid_Size_X"]) for r in fl)
    l0 = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in fl if int(r["Grid_Size_X"]) == gmax]
    print(f"level-0 attention ({[r for r in fl if int(r['Grid_Size_X']) == gmax][0]['Kernel_Name'].split('(')[0][-40:]}, grid {gmax // 256} workgroups): {len(l0)} launches, "
          f"avg {sum(l0) / len(l0):.3f} ms, min {min(l0):.3f}, max {max(l0):.3f}")
if len(sys.argv) > 3:
    ig = [r for r in rows if sys.argv[3] in r["Kernel_Name"]]
    n = len(ig) // n_fwd
    for r in ig[-n:]:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(r["Kernel_Name"][40:64], "grid", int(r["Grid_Size_X"]) // 256, r["Grid_Size_Y"], r["Grid_Size_Z"], f"{us:9.1f} us")
