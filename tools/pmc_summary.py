"""Per-kernel HBM-side traffic of the bench workload from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).
usage: pmc_summary.py <fetch_dir> <write_dir> <n_forwards>.  FETCH_SIZE is doubled (gfx950 tallies 128-B requests
at 64 B); both counters are in KiB.  Durations are the dispatch timestamps of the same (profiled) runs."""
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


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
n = int(sys.argv[3])
print(f"{'kernel':58s} {'calls/fwd':>9s} {'ms/fwd':>8s} {'read GB':>8s} {'write GB':>8s} {'GB/s':>7s}   (per forward; read = FETCH_SIZE x2)")
tot = [0.0, 0.0, 0.0]
for k in sorted(fe, key=lambda k: -fe[k][2]):
    c, f_kib, ms = fe[k]
    w_kib = wr.get(k, [0, 0.0, 0.0])[1]
    rd, wrb = 2 * f_kib * 1024 / 1e9 / n, w_kib * 1024 / 1e9 / n
    ms /= n
    tot[0] += rd; tot[1] += wrb; tot[2] += ms
    if ms < 0.05:
        continue
    print(f"{k[:58]:58s} {c / n:9.1f} {ms:8.3f} {rd:8.3f} {wrb:8.3f} {(rd + wrb) / (ms / 1e3):7.0f}")
print(f"{'all kernels':58s} {'':9s} {tot[2]:8.3f} {tot[0]:8.3f} {tot[1]:8.3f} {(tot[0] + tot[1]) / (tot[2] / 1e3):7.0f}")
