"""Full-size (256x256, attention over all 65 536 tokens) forward of the JPEG model computed by the numpy ORACLE.

The reference itself cannot run this size here (its attention materialises N x N = 64 GiB per image at level 0), so
this fixture is produced by oracle/ddpm_oracle.py -- which is pinned against the reference's own outputs at 32x32 /
64x64 / 96x64 by the other fixtures of this directory -- and lets the GPU test compare the fp32 HIP path at
BASELINE.json's image size without a 15 s CPU run inside the test.  Usage: python tests/golden/make_fullsize_oracle.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ddpm_image_restoration_amd import spec, synth  # noqa: E402
from oracle import ddpm_oracle as O  # noqa: E402

SEED_W, SEED_X = 0, 31
x = synth.synth_images(1, 256, seed=SEED_X)
t = np.array([0.43], np.float32)
cl = np.array([0.9], np.float32)
m = O.Model(spec.ARCH_JPEG, synth.synth_state_dict(spec.ARCH_JPEG, SEED_W))
t0 = time.time()
y = m.forward(x, t, cl)
print(f"oracle forward 256x256: {time.time() - t0:.1f} s, |y| mean {np.abs(y).mean():.4f}")
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg_fwd256_oracle.npz")
np.savez_compressed(out, y=y.astype(np.float32), t=t, cl=cl, seed_w=SEED_W, seed_x=SEED_X)
print("wrote", out, os.path.getsize(out), "bytes")
