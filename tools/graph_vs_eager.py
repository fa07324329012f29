"""Timestep launch: eager vs captured hipGraphs (use_graph), ms per reverse-diffusion step of the JPEG family with the
device codec, at config-1 size (32x32, B=4), 64x64 and the bench size (256x256, B=16); both arithmetic modes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import codec, spec, synth  # noqa: E402
from ddpm_image_restoration_amd.model import JPEGDiffusionModel  # noqa: E402
from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler  # noqa: E402

dev = torch.device("cuda", 0)
sd = synth.synth_state_dict(spec.ARCH_JPEG)
for dtype in ("bf16", "f32"):
    models = {}
    for g in (False, True):
        m = JPEGDiffusionModel(device=dev, dtype=dtype, use_graph=g)
        m.load_state_dict(sd)
        models[g] = DDRMJPEGSampler(m)
    for R, B, K in ((32, 4, 200), (64, 4, 200), (256, 16, 10 if dtype == "bf16" else 3)):
        y = codec.jpeg_compress(torch.from_numpy(synth.synth_images(B, R, seed=1)), 10).to(dev)
        res = {}
        for rep in range(2):   # interleaved: eager, graph, eager, graph
            for g in (False, True):
                s = models[g]
                s.sample(y, 10, steps=1000, start_step=999, num_steps=3, seed=1)            # warm / capture
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = s.sample(y, 10, steps=1000, start_step=996, num_steps=K, seed=1)
                torch.cuda.synchronize()
                res.setdefault(g, []).append((time.perf_counter() - t0) * 1e3 / K)
                res[("out", g)] = out
        same = torch.equal(res[("out", False)], res[("out", True)])
        print(f"{dtype} {R}x{R} B={B}: eager {min(res[False]):.3f} ms/step, graph {min(res[True]):.3f} ms/step, bit-equal {same}", flush=True)
