"""Reference forward at 128 x 128 (attention over 16 384 tokens at level 0) -> tests/golden/{jpeg,webp,avif}_fwd128_ref.npz.

Same mechanism as make_golden.py (the reference's own source is executed by line range from /root/reference with the
synthetic weights of ddpm_image_restoration_amd.synth); kept separate because the reference materialises the N x N
attention matrix (4 heads x 1 GiB at this size, 8 heads for the AVIF model) and takes about a minute per model on 8 cores.  256 x 256 is out of its
reach here (64 GiB of scores), which is what tests/golden/make_fullsize_oracle.py is for.
Usage: python tests/golden/make_golden_128.py [jpeg] [webp] [avif]      (default: the families whose file is missing)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
from ddpm_image_restoration_amd import spec, synth  # noqa: E402


def main():
    import torch
    torch.manual_seed(0)
    todo = sys.argv[1:] or [nm for nm in ("jpeg", "webp", "avif") if not os.path.exists(os.path.join(HERE, f"{nm}_fwd128_ref.npz"))]
    for arch, nm in ((spec.ARCH_JPEG, "jpeg"), (spec.ARCH_WEBP, "webp"), (spec.ARCH_AVIF, "avif")):
        if nm not in todo:
            continue
        ns, m, _Sampler, _comp = G.build_model(arch)
        x = synth.synth_images(1, 128, seed=51)
        t = np.array([0.62], dtype=np.float32)
        cl = np.array([0.2], dtype=np.float32)
        t0 = time.time()
        with torch.no_grad():
            y = m(torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(cl)).numpy()
        print(f"{nm}: reference forward 128x128 in {time.time() - t0:.1f} s, |y| mean {np.abs(y).mean():.4f}")
        np.savez_compressed(os.path.join(HERE, f"{nm}_fwd128_ref.npz"), y=y.astype(np.float32), t=t, cl=cl,
                            seed_w=np.int32(0), seed_x=np.int32(51))
        del m


if __name__ == "__main__":
    main()
