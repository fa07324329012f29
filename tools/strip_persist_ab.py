"""A/B of the tile-walking strip kernel (experiments build: DDPMIR_NO_PERSIST = one workgroup per tile again), interleaved child processes."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys; sys.path.insert(0, %r); from ddpm_image_restoration_amd import ops\n"
        "for (b, h, ci, co) in ((8, 512, 64, 64), (8, 512, 128, 64), (16, 256, 64, 64), (8, 256, 256, 64), (16, 128, 64, 64)):\n"
        "    print(b, h, ci, co, min(ops.conv_time(b, h, h, ci, co, 3, 'dma', True, False, 0, 5) for _ in range(3)) * 1e3)\n" % root)
res = {}
for rnd in range(2):
    for sw in ("", "DDPMIR_NO_PERSIST"):
        env = dict(os.environ, DDPMIR_LIB_PATH=os.path.join(root, "ddpm_image_restoration_amd", "libddpmir_exp.so"))
        if sw:
            env[sw] = "1"
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout
        for ln in out.splitlines():
            f = ln.split()
            if len(f) == 5:
                res.setdefault((tuple(f[:4]), sw), []).append(float(f[4]))
for (shape, sw), v in sorted(res.items()):
    print(f"B={shape[0]:>2s} {shape[1]:>3s}x{shape[1]:<3s} {shape[2]:>3s}->{shape[3]:<3s} [{sw or 'tile walk (shipped)':22s}] us: {['%.1f' % x for x in v]}", flush=True)
