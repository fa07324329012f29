"""Full-size forwards computed by the numpy ORACLE for the configurations the reference cannot run here.

The reference materialises the N x N attention matrix (64 GiB per image at level 0 of a 256 x 256 input), so for
BASELINE.json's image sizes the fixtures are produced by oracle/ddpm_oracle.py -- which is itself pinned against the
reference's own outputs at 32x32 / 64x64 / 96x64 / 128x128 by the other fixtures of this directory.  They let the GPU
tests compare the fp32 HIP path at full size without minutes of CPU work inside the test.

Cases (name -> family, size, attention levels):
  jpeg_fwd256          BASELINE configs[1]   JPEG U-Net, 256 x 256, attention in all 13 blocks
  webp_fwd256          BASELINE configs[2]   WebP U-Net, 256 x 256, attention in all 13 blocks
  avif_fwd128/256      BASELINE configs[3]   AVIF U-Net (8 heads: head_dim 8 at level 0), 128 x 128 and 256 x 256
  jpeg_fwd512_bott     BASELINE configs[4]   JPEG U-Net, 512 x 512, attention in the bottleneck blocks only (attn_min_level=5)
The 512 x 512 output is stored on a 2 x 2 pixel stride (every output pixel depends on the whole image through the
bottleneck attention, so the subset pins the whole computation) plus float64 checksums of the full tensor.

Usage: python tests/golden/make_fullsize_oracle.py [case ...]        (default: every case whose file is missing)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ddpm_image_restoration_amd import spec, synth  # noqa: E402
from oracle import ddpm_oracle as O  # noqa: E402

SEED_W = 0
#        name               arch            size  attn_min_level  seed_x  t      cl    stride
CASES = {
    "jpeg_fwd256_oracle": (spec.ARCH_JPEG, 256, 0, 31, 0.43, 0.9, 1),
    "webp_fwd256_oracle": (spec.ARCH_WEBP, 256, 0, 33, 0.71, 0.71, 1),
    "avif_fwd128_oracle": (spec.ARCH_AVIF, 128, 0, 34, 0.55, 0.3, 1),
    "avif_fwd256_oracle": (spec.ARCH_AVIF, 256, 0, 35, 0.18, 0.18, 1),
    "jpeg_fwd512_bott_oracle": (spec.ARCH_JPEG, 512, 5, 36, 0.36, 0.36, 2),
}


def make(name):
    arch, size, aml, seed_x, t, cl, stride = CASES[name]
    x = synth.synth_images(1, size, seed=seed_x)
    t = np.array([t], np.float32)
    cl = np.array([cl], np.float32)
    m = O.Model(arch, synth.synth_state_dict(arch, SEED_W), attn_min_level=aml)
    t0 = time.time()
    y = m.forward(x, t, cl).astype(np.float32)
    print(f"{name}: oracle forward {size}x{size} in {time.time() - t0:.1f} s, |y| mean {np.abs(y).mean():.4f}", flush=True)
    out = os.path.join(HERE, name + ".npz")
    np.savez_compressed(out, y=y[:, :, ::stride, ::stride], t=t, cl=cl, seed_w=SEED_W, seed_x=seed_x,
                        stride=np.int32(stride), attn_min_level=np.int32(aml),
                        sum=np.float64(y.astype(np.float64).sum()), abs_sum=np.float64(np.abs(y.astype(np.float64)).sum()))
    print("wrote", out, os.path.getsize(out), "bytes", flush=True)


if __name__ == "__main__":
    names = sys.argv[1:] or [n for n in CASES if not os.path.exists(os.path.join(HERE, n + ".npz"))]
    for n in names:
        make(n)
