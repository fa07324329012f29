"""A second, independent set of reference forwards: weights from synth seed 1, other images and (t, compression_level) pairs,
all three families at 32 x 32 and 64 x 32 -> tests/golden/seed1_fwd.npz.  Same mechanism as make_golden.py (the reference's
own source executed by line range from /root/reference).  Usage: python tests/golden/make_golden_seed1.py"""
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
    out = {}
    for arch, nm in ((spec.ARCH_WEBP, "webp"), (spec.ARCH_JPEG, "jpeg"), (spec.ARCH_AVIF, "avif")):
        ns, m, _S, _c = G.build_model(arch, seed=1)
        x = synth.synth_images(3, 32, seed=201)
        t = np.array([0.02, 0.5, 0.97], np.float32)
        cl = np.array([0.9, 0.5, 0.0], np.float32)
        x2 = synth.synth_images(1, 64, seed=202)[:, :, :, :32].copy()          # 64 rows x 32 columns
        t2 = np.array([0.33], np.float32)
        with torch.no_grad():
            y = m(torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(cl)).numpy()
            y2 = m(torch.from_numpy(x2), torch.from_numpy(t2), torch.from_numpy(t2)).numpy()
        out.update({f"{nm}_x": x, f"{nm}_t": t, f"{nm}_cl": cl, f"{nm}_y": y, f"{nm}_x2": x2, f"{nm}_t2": t2, f"{nm}_y2": y2})
        print(nm, float(np.abs(y).mean()), float(np.abs(y2).mean()))
    p = os.path.join(HERE, "seed1_fwd.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p))


if __name__ == "__main__":
    main()
