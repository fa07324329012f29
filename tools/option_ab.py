"""Same-process A/B of a 0 / 1 context option (epilogue_direct, head_fused, ...): ms per timestep of configs[4]'s shape in three modes and
of the headline forward without level-0 attention, interleaved, min of three.   usage: option_ab.py <option> [value_a value_b]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import codec, spec, synth  # noqa: E402
from ddpm_image_restoration_amd.model import JPEGDiffusionModel  # noqa: E402
from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler  # noqa: E402

opt = sys.argv[1]
va, vb = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1.0)
dev = torch.device("cuda", 0)
sd = synth.synth_state_dict(spec.ARCH_JPEG)
for name, kw, R, B, q in (("configs[4] bf16", dict(dtype="bf16", attn_levels="bottleneck"), 512, 8, 20), ("configs[4] fp8", dict(dtype="fp8", attn_levels="bottleneck"), 512, 8, 20),
                          ("configs[4] fp8@64", dict(dtype="fp8", attn_levels="bottleneck", fp8_min_channels=64), 512, 8, 20),
                          ("headline-minus-l0 (256, B=16, bottleneck attention)", dict(dtype="bf16", attn_levels="bottleneck"), 256, 16, 10)):
    m = JPEGDiffusionModel(device=dev, **kw)
    m.load_state_dict(sd)
    y = codec.jpeg_compress(torch.from_numpy(synth.synth_images(B, R, seed=3)), q).to(dev)
    s = DDRMJPEGSampler(m)
    best = {}
    for rep in range(3):
        for v in (va, vb):
            m.set_option(opt, v)
            s.sample(y, q, steps=1000, start_step=999, num_steps=2, seed=1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.sample(y, q, steps=1000, start_step=997, num_steps=10, seed=1)
            torch.cuda.synchronize()
            best[v] = min(best.get(v, 1e9), (time.perf_counter() - t0) * 100)
    print(f"{name}: ms/step  {opt} = {va:g}: {best[va]:.3f} | {opt} = {vb:g}: {best[vb]:.3f}", flush=True)
    del s, m
    torch.cuda.empty_cache()
