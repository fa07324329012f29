"""Restored-PSNR difference of the reduced-precision modes against fp32 parity mode at the benchmarked size, over more images
than the -m gpu test affords (tests/test_gpu_psnr256.py: 16 images; its standard error of the mean is ~0.0035 dB).
usage: python tools/psnr256_bias.py [n_batches=4]   -> JPEG, 256 x 256, q10, 80 steps, 16 images per batch, new images and noise
seed per batch; modes: bf16 (default fp32 stream), bf16 + bf16 activation stream, fp8."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import harness, spec, synth
from ddpm_image_restoration_amd.model import MODEL_BY_ARCH
from ddpm_image_restoration_amd.sampler import SAMPLER_BY_ARCH

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
arch, R, q = spec.ARCH_JPEG, 256, 10


def psnr_each(a, x0):
    a01, b01 = np.clip(a * 0.5 + 0.5, 0, 1), np.clip(x0 * 0.5 + 0.5, 0, 1)
    return -10 * np.log10(((a01 - b01) ** 2).reshape(a.shape[0], -1).mean(axis=1) + 1e-8)


def model(dtype, stream=None):
    m = MODEL_BY_ARCH[arch](device=torch.device("cuda", 0), dtype=dtype)
    m.load_state_dict(STATE)
    if stream is not None:
        m.set_option("bf16_activation_stream", stream)
    return m


STATE = synth.synth_state_dict(arch, 0)   # the seed-0 synthetic weights of the test-suite
steps = harness.steps_for_quality(arch, q)
modes = [("f32", "f32", None), ("bf16", "bf16", None), ("bf16+stream16", "bf16", 1), ("fp8", "fp8", None)]
diffs = {n: [] for n, _, _ in modes[1:]}
for bi in range(nb):
    x0 = synth.synth_images(16, R, seed=1000 + bi)
    y = harness.COMPRESS[arch](torch.from_numpy(x0), q).cuda()
    ps = {}
    for name, dtype, stream in modes:
        m = model(dtype, stream)
        ps[name] = psnr_each(SAMPLER_BY_ARCH[arch](m).sample(y, q, steps=steps, seed=3 + bi).cpu().numpy(), x0)
        del m
        torch.cuda.empty_cache()
    for n in diffs:
        diffs[n].append(ps[n] - ps["f32"])
    print(f"batch {bi}: f32 {ps['f32'].mean():.4f} dB; " + "; ".join(f"{n} {np.mean(ps[n] - ps['f32']):+.5f}" for n in diffs), flush=True)
for n, d in diffs.items():
    d = np.concatenate(d)
    print(f"{n:16s} - f32 over {d.size} images: mean {d.mean():+.5f} dB +- {d.std(ddof=1) / np.sqrt(d.size):.5f} (standard error), "
          f"std {d.std():.4f}, per-image max {np.abs(d).max():.4f}")
