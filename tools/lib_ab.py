"""A/B two builds of libddpmir.so (tools/ab/*.so) on one device: interleaved rounds of the level-0 attention."""
import sys, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or ["lib_base.so", "lib_pp.so"]
code = "import os,sys; sys.path.insert(0,'.'); from ddpm_image_restoration_amd import ops; print(min(ops.attention_time(4,65536,64,4,1,3) for _ in range(2)))"
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, DDPMIR_LIB_PATH=os.path.join(root, "tools", "ab", l))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=root).stdout.strip().split("\n")[-1]
        res[l].append(float(out))
for l in libs:
    print(f"{l}: min {min(res[l]):.2f} ms  all {['%.2f' % x for x in res[l]]}", flush=True)
