"""A/B two builds of the library (tools/ab/*.so, selected through DDPMIR_LIB_PATH) on ONE device: interleaved child
processes running the default bench slice and, optionally, the GEMM inventory.  usage: lib_ab_bench.py a.so b.so [--gemm]"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a for a in sys.argv[1:] if not a.startswith("--")]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, DDPMIR_LIB_PATH=os.path.join(root, "tools", "ab", l))
        out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--no-cpu-baseline"], env=env, capture_output=True, text=True, cwd=root).stdout
        d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
        r = d["roofline"]
        res[l].append((d["ms_per_step"], r["forward_ms"] - 2 * r["avg_launch_ms"]))
for l in libs:
    print(f"{l}: ms/step {['%.2f' % a for a, _ in res[l]]}  forward minus level-0 attention {['%.2f' % b for _, b in res[l]]}", flush=True)
if "--gemm" in sys.argv:
    for l in libs:
        env = dict(os.environ, DDPMIR_LIB_PATH=os.path.join(root, "tools", "ab", l))
        out = subprocess.run([sys.executable, "tools/gemm_inventory.py"], env=env, capture_output=True, text=True, cwd=root).stdout
        print(l, out.strip().splitlines()[-1], flush=True)
