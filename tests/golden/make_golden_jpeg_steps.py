"""Every step i = 49 ... 39 of the 50-step JPEG q10 reference run of make_golden.py -> tests/golden/jpeg_steps49_39.npz.

jpeg.npz keeps the trajectory at i = 49, 40, 30, ...; one of the four images leaves the reference trajectory between the
first two (a uint8 truncation that lands on the other side of an integer).  This fixture holds x_t fed to
the model and x_theta for every step in between, so the GPU test can name the step at which that happens and check that
up to it the HIP path tracks the reference.  Same mechanism as make_golden.py: the reference's own source executed by
line range from /root/reference.   Usage: python tests/golden/make_golden_jpeg_steps.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
from ddpm_image_restoration_amd import spec, synth  # noqa: E402


def main():
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ns, m, Sampler, comp = G.build_model(spec.ARCH_JPEG)
    g = np.load(os.path.join(HERE, "jpeg.npz"))
    q, steps, B = int(g["samp_q"]), int(g["samp_steps"]), g["samp_y"].shape[0]
    noise = synth.synth_noise(steps, B, 32, 32, seed=7)
    xs, thetas = [], []
    def rec(_m, i, o):
        xs.append(i[0].numpy().copy())
        thetas.append(o.numpy().copy())

    hk = m.register_forward_hook(rec)
    final = G.run_sampler(ns, m, Sampler, g["samp_y"], q, steps, noise)
    hk.remove()
    assert np.array_equal(final, g["samp_final"]), "the reference run is not reproducible?"
    out = {}
    for i in range(49, 38, -1):
        out[f"xt_in_{i}"] = xs[steps - 1 - i]
        out[f"theta_{i}"] = thetas[steps - 1 - i]
    for i in (49, 40):
        assert np.array_equal(out[f"xt_in_{i}"], g[f"samp_xt_in_{i}"]) and np.array_equal(out[f"theta_{i}"], g[f"samp_theta_{i}"])
    p = os.path.join(HERE, "jpeg_steps49_39.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
