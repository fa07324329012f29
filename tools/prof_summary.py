import csv, sys, collections, glob
d = sys.argv[1]
t = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(t)))
n_fwd = int(sys.argv[2]) if len(sys.argv) > 2 else 3
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r["Kernel_Name"].replace("void ", "").replace("ddpmir::(anonymous namespace)::", "").split("(")[0]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[k][0] += 1; agg[k][1] += us
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot/1e3:.2f} ms over {n_fwd} forwards = {tot/1e3/n_fwd:.2f} ms/forward")
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{k[:60]:60s} calls/fwd={c/n_fwd:6.1f} ms/fwd={us/1e3/n_fwd:8.3f}")
if len(sys.argv) > 3:
    ig = [r for r in rows if sys.argv[3] in r["Kernel_Name"]]
    n = len(ig) // n_fwd
    for r in ig[-n:]:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(r["Kernel_Name"][40:64], "grid", int(r["Grid_Size_X"]) // 256, r["Grid_Size_Y"], r["Grid_Size_Z"], f"{us:9.1f} us")
