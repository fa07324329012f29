"""Merged two-lane schedule (levels below full resolution once per timestep) against fully separate lanes, whole 80-step
restorations at 256 x 256 in fp32 mode: images are independent, so the two schedules must agree bit for bit.
Runs each schedule in a child process (DDPMIR_SEPARATE_LANES exists in the experiments build only).
usage: python tools/lanes_check.py"""
import os, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from ddpm_image_restoration_amd import harness, spec, synth
from ddpm_image_restoration_amd.model import MODEL_BY_ARCH
from ddpm_image_restoration_amd.sampler import SAMPLER_BY_ARCH
arch = int(sys.argv[1])
m = MODEL_BY_ARCH[arch](device=torch.device("cuda", 0)); m.load_state_dict(synth.synth_state_dict(arch))
x0 = synth.synth_images(4, 256, seed=77)
y = harness.COMPRESS[arch](torch.from_numpy(x0), 10).cuda()
out = SAMPLER_BY_ARCH[arch](m).sample(y, 10, steps=harness.steps_for_quality(arch, 10), seed=3).cpu().numpy()
np.save(sys.argv[2], out)
''' % root
exp = os.path.join(root, "ddpm_image_restoration_amd", "libddpmir_exp.so")
import numpy as np
for arch, name in ((0, "webp"), (2, "avif")):
    outs = []
    for sep in (0, 1):
        f = tempfile.mktemp(suffix=".npy")
        env = dict(os.environ, DDPMIR_LIB_PATH=exp)
        if sep: env["DDPMIR_SEPARATE_LANES"] = "1"
        subprocess.run([sys.executable, "-c", code, str(arch), f], env=env, check=True)
        outs.append(np.load(f)); os.remove(f)
    d = np.abs(outs[0] - outs[1])
    print(f"{name}: merged vs separate lanes, 4 images, full schedule, fp32 mode: max |diff| {d.max():.3e}, equal: {np.array_equal(outs[0], outs[1])}, finite: {np.isfinite(outs[0]).all()}", flush=True)
