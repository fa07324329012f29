"""configs[4] (512x512, attention at the bottleneck only, JPEG q20): fp8 mode (e4m3 operands in the 3x3 convs with >= 256
channels) against bf16 mode -- forward error vs fp32 mode, restored-PSNR difference, and per-shape GEMM times bf16 vs fp8."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import codec, ops, spec, synth  # noqa: E402
from ddpm_image_restoration_amd.model import JPEGDiffusionModel  # noqa: E402
from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler  # noqa: E402

dev = torch.device("cuda", 0)
sd = synth.synth_state_dict(spec.ARCH_JPEG)
print("== GEMM shapes of the deep 3x3 convs at 512x512, B=8 (ms per call: bf16 | fp8)")
for (h, ci, co) in ((512, 64, 64), (512, 128, 64), (256, 128, 128), (256, 256, 64), (128, 256, 256), (64, 256, 512), (64, 512, 512), (64, 1024, 256), (32, 512, 512), (32, 1024, 512), (16, 512, 1024), (16, 1024, 1024), (16, 1024, 512)):
    a = min(ops.conv_time(8, h, h, ci, co, 3, "dma", True, False, 0, 5) for _ in range(2))
    b = min(ops.conv_time(8, h, h, ci, co, 3, "fp8", True, False, 0, 5) for _ in range(2))
    fl = 2.0 * 8 * h * h * ci * co * 9
    print(f"H={h:3d} {ci:4d}->{co:4d}: bf16 {a * 1e3:7.1f} us ({fl / a / 1e9:6.1f} TF/s)   fp8 {b * 1e3:7.1f} us ({fl / b / 1e9:6.1f} TF/s)   x{a / b:.2f}", flush=True)

models = {}
for dt in ("f32", "bf16", "fp8", "fp8@128", "fp8@64"):
    m = JPEGDiffusionModel(device=dev, dtype=dt.split("@")[0], attn_levels="bottleneck",
                           fp8_min_channels=int(dt.split("@")[1]) if "@" in dt else None)
    m.load_state_dict(sd)
    models[dt] = m
for R, B in ((128, 2), (256, 2), (512, 1)):
    x = torch.from_numpy(synth.synth_images(B, R, seed=70 + R)).to(dev)
    t = torch.full((B,), 0.4, device=dev)
    y = {dt: models[dt](x, t, t) for dt in models}
    for dt in [k for k in models if k != "f32"]:
        e = (y[dt] - y["f32"]).abs()
        print(f"forward {R}x{R}: {dt} vs f32  max {float(e.max()):.4f} mean {float(e.mean()):.5f}", flush=True)
print("== restored PSNR, JPEG q20, 64 steps schedule (clamp(100-q)), 64x64, 64 images")
x0 = synth.synth_images(64, 64, seed=777)
yq = codec.jpeg_compress(torch.from_numpy(x0), 20).to(dev)


def psnr_each(a):
    a01, b01 = np.clip(a * 0.5 + 0.5, 0, 1), np.clip(x0 * 0.5 + 0.5, 0, 1)
    return -10 * np.log10(((a01 - b01) ** 2).reshape(a.shape[0], -1).mean(axis=1) + 1e-8)


res = {dt: psnr_each(DDRMJPEGSampler(models[dt]).sample(yq, 20, steps=80, seed=5).cpu().numpy()) for dt in models}
for dt in [k for k in models if k != "f32"]:
    d = res[dt] - res["f32"]
    print(f"{dt}: mean restored PSNR {res[dt].mean():.4f} dB, minus f32 {d.mean():+.4f} dB (std {d.std():.4f}, max |d| {np.abs(d).max():.3f})", flush=True)
print("== ms per timestep, 512x512, B=8, q20, T=1000 slice")
y512 = codec.jpeg_compress(torch.from_numpy(synth.synth_images(8, 512, seed=3)), 20).to(dev)
for rep in range(2):
    for dt in [k for k in models if k != "f32"]:
        s = DDRMJPEGSampler(models[dt])
        s.sample(y512, 20, steps=1000, start_step=999, num_steps=2, seed=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.sample(y512, 20, steps=1000, start_step=997, num_steps=10, seed=1)
        torch.cuda.synchronize()
        print(f"{dt}: {(time.perf_counter() - t0) * 100:.3f} ms/step", flush=True)
