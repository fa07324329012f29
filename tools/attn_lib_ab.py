"""Level-0 attention launch (B=16, N=65536, C=64, 4 heads, random operands: the no-max kernel) of several builds of the library on ONE
device, interleaved child processes calling ddpmir_op_attention_time through ctypes.  usage: attn_lib_ab.py a.so b.so ... (tools/ab/)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
code = ("import ctypes as C,sys; l=C.CDLL(sys.argv[1]); ms=C.c_float(0); "
        "rc=l.ddpmir_op_attention_time(16,65536,64,4,1,6,C.byref(ms)); print(rc, ms.value)")
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", code, os.path.join(root, "tools", "ab", l)], capture_output=True, text=True).stdout.split()
        res[l].append(float(out[1]) if len(out) == 2 and out[0] == "0" else float("nan"))
for l in libs:
    print(f"{l:22s} ms per level-0 attention call (pack + bounds + flash): {['%.3f' % v for v in res[l]]}", flush=True)
