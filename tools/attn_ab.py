"""A/B attention variants in ONE process on one device (interleaved rounds), level-0 shape."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the variant is read once per process (static), so each variant runs in a child; rounds interleave children
variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "5", "7", "8", "9"]
B = int(os.environ.get("AB_BATCH", "4"))
HEADS = int(os.environ.get("AB_HEADS", "4"))
code = f"import os,sys; sys.path.insert(0,'.'); from ddpm_image_restoration_amd import ops; print(min(ops.attention_time({B},65536,64,{HEADS},1,3) for _ in range(2)))"
res = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        env = dict(os.environ, DDPMIR_ATTN_VARIANT=v)
        env.setdefault("DDPMIR_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ddpm_image_restoration_amd", "libddpmir_exp.so"))   # variants exist in the experiments build only
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
        res[v].append(float(out))
for v in variants:
    print(f"variant {v}: min {min(res[v]):.2f} ms  all {['%.2f' % x for x in res[v]]}", flush=True)
