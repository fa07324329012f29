"""Same-process A/B of the 256 x 256 eight-wave tile (csrc/gemm256.hip) against the 128-row kernel it replaces:
(1) per-shape times of the Cout >= 256 3x3 convs of the 256 x 256 (B=16) and 512 x 512 (B=8) pyramids, bf16 and e4m3;
(2) ms per timestep of configs[4] (bf16 / fp8 modes) and of the headline forward with option gemm256_tile = 0 / -1 / 1,
interleaved, min of `rounds`."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import codec, ops, spec, synth  # noqa: E402
from ddpm_image_restoration_amd.model import JPEGDiffusionModel  # noqa: E402
from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = torch.device("cuda", 0)
if what in ("all", "shapes"):
    print("== 3x3 convs with Cout >= 256: us per launch (TFLOP/s), 128-row kernel | 256 tile forced; fp32 + bf16 outputs")
    shapes = [(8, 128, 128, 256), (8, 128, 256, 256), (8, 64, 256, 512), (8, 64, 512, 512), (8, 64, 1024, 256), (8, 64, 256, 256),
              (8, 32, 512, 512), (8, 32, 1024, 512), (8, 16, 512, 1024), (8, 16, 1024, 1024), (8, 16, 1024, 512),
              (16, 64, 128, 256), (16, 64, 256, 256), (16, 32, 256, 512), (16, 32, 512, 512), (16, 32, 1024, 256),
              (16, 16, 512, 512), (16, 16, 1024, 512), (16, 8, 512, 1024), (16, 8, 1024, 1024)]
    for mode in ("dma", "fp8"):
        for (b, h, ci, co) in shapes:
            if mode == "fp8" and ci % 128:
                continue
            t = {}
            for tile in (0, 1):
                ops.set_tile256(tile)
                t[tile] = min(ops.conv_time(b, h, h, ci, co, 3, mode, True, False, 0, 5) for _ in range(3))
            fl = 2.0 * b * h * h * ci * co * 9
            print(f"{mode} B={b:2d} {h:3d}x{h:<3d} {ci:4d}->{co:4d}: {t[0] * 1e3:7.1f} us ({fl / t[0] / 1e9:6.0f})  |  {t[1] * 1e3:7.1f} us ({fl / t[1] / 1e9:6.0f})   x{t[0] / t[1]:.2f}", flush=True)
    ops.set_tile256(-1)

if what in ("all", "model"):
    sd = synth.synth_state_dict(spec.ARCH_JPEG)
    rounds = 3
    cases = [("configs[4] bf16", dict(dtype="bf16", attn_levels="bottleneck"), 512, 8, 20),
             ("configs[4] fp8", dict(dtype="fp8", attn_levels="bottleneck"), 512, 8, 20),
             ("configs[4] fp8@64", dict(dtype="fp8", attn_levels="bottleneck", fp8_min_channels=64), 512, 8, 20),
             ("headline-minus-l0 (256, B=16, bottleneck attention)", dict(dtype="bf16", attn_levels="bottleneck"), 256, 16, 10)]
    for name, kw, R, B, q in cases:
        m = JPEGDiffusionModel(device=dev, **kw)
        m.load_state_dict(sd)
        y = codec.jpeg_compress(torch.from_numpy(synth.synth_images(B, R, seed=3)), q).to(dev)
        s = DDRMJPEGSampler(m)
        best = {}
        for rep in range(rounds):
            for opt in (0, -1, 1):
                m.set_option("gemm256_tile", opt)
                s.sample(y, q, steps=1000, start_step=999, num_steps=2, seed=1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s.sample(y, q, steps=1000, start_step=997, num_steps=10, seed=1)
                torch.cuda.synchronize()
                best[opt] = min(best.get(opt, 1e9), (time.perf_counter() - t0) * 100)
        print(f"{name}: ms/step  tile off {best[0]:.3f} | auto {best[-1]:.3f} | forced {best[1]:.3f}", flush=True)
        del s, m
        torch.cuda.empty_cache()
