"""Same-process A/B of the register-direct epilogue of the 64-column GEMM kernels (option epilogue_direct): per-shape launch times
(fp32 + bf16 outputs / bf16 only / fp32 only) and ms per timestep of configs[4]'s shape and of the headline forward without level-0 attention."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import codec, ops, spec, synth  # noqa: E402
from ddpm_image_restoration_amd.model import JPEGDiffusionModel  # noqa: E402
from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler  # noqa: E402

dev = torch.device("cuda", 0)
what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("all", "shapes"):
    for (b, h, ci, co, k) in ((8, 512, 8, 64, 3), (8, 512, 64, 64, 3), (8, 512, 128, 64, 3), (16, 256, 64, 64, 3), (16, 256, 128, 64, 3), (8, 512, 64, 64, 1), (8, 256, 256, 64, 3)):
        row = []
        for name, dbg, gelu in (("fp32+bf16", 0, True), ("bf16 only", 128, True), ("fp32 only", 0, False)):
            t = {}
            for d in (0, 1):
                ops.set_epilogue_direct(d)
                t[d] = min(ops.conv_time(b, h, h, ci, co, k, "dma", gelu, False, dbg, 5) for _ in range(3)) * 1e3
            row.append(f"{name}: {t[0]:6.1f} -> {t[1]:6.1f} us")
        print(f"B={b} {h}x{h} {ci:3d}->{co:3d} k{k}:  " + " | ".join(row), flush=True)
    ops.set_epilogue_direct(1)
if what in ("all", "model"):
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
            for opt in (0, 1):
                m.set_option("epilogue_direct", opt)
                s.sample(y, q, steps=1000, start_step=999, num_steps=2, seed=1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s.sample(y, q, steps=1000, start_step=997, num_steps=10, seed=1)
                torch.cuda.synchronize()
                best[opt] = min(best.get(opt, 1e9), (time.perf_counter() - t0) * 100)
        print(f"{name}: ms/step  LDS epilogue {best[0]:.3f} | register-direct {best[1]:.3f}", flush=True)
        del s, m
        torch.cuda.empty_cache()
