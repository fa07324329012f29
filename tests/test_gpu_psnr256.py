"""north_star: "identical PSNR to 2 decimals" -- for the arithmetic the benchmark runs (bf16 operands) at the benchmarked
size.  16 images at 256 x 256, JPEG and WebP q10, the harness' 80-step schedule (webp_inference.py:678-679), device noise:
restored PSNR of bf16 mode against fp32 parity mode (which tracks the reference to < 1e-3 per pixel, test_gpu_model.py).

The sampling loop is chaotic (uint8 truncation + a lossy codec every step), so two arithmetic modes do not
restore the same pixels; what is equal is the PSNR: the per-image difference is a zero-mean scatter (measured, round 2:
JPEG mean +0.001 dB, std 0.015, max 0.038; WebP +0.003 / 0.015 / 0.038).  Bars: |mean| < 0.01 dB, per image < 0.1 dB.
"""
import numpy as np
import pytest

from ddpm_image_restoration_amd import spec, synth

pytestmark = pytest.mark.gpu


def psnr_each(a, x0):
    a01, b01 = np.clip(a * 0.5 + 0.5, 0, 1), np.clip(x0 * 0.5 + 0.5, 0, 1)
    return -10 * np.log10(((a01 - b01) ** 2).reshape(a.shape[0], -1).mean(axis=1) + 1e-8)


@pytest.mark.parametrize("arch", [spec.ARCH_JPEG, spec.ARCH_WEBP])
def test_restored_psnr_bf16_equals_fp32_mode_at_256(arch, synth_state):
    import torch
    from ddpm_image_restoration_amd import harness
    from ddpm_image_restoration_amd.model import MODEL_BY_ARCH
    from ddpm_image_restoration_amd.sampler import SAMPLER_BY_ARCH
    R, n_img, q = 256, 16, 10
    x0 = synth.synth_images(n_img, R, seed=777)
    y = harness.COMPRESS[arch](torch.from_numpy(x0), q).cuda()
    steps = harness.steps_for_quality(arch, q)
    assert steps == 80
    res = {}
    for mode in ("f32", "bf16"):
        m = MODEL_BY_ARCH[arch](device=torch.device("cuda", 0), dtype=mode)
        m.load_state_dict(synth_state(arch))
        res[mode] = SAMPLER_BY_ARCH[arch](m).sample(y, q, steps=steps, seed=3).cpu().numpy()
        del m
        torch.cuda.empty_cache()
    p32, p16 = psnr_each(res["f32"], x0), psnr_each(res["bf16"], x0)
    d = p16 - p32
    name = {v: k for k, v in spec.ARCH_NAMES.items()}[arch]
    print(f"{name}: restored fp32 {p32.mean():.4f} dB; bf16 - fp32 mean {d.mean():+.5f} dB, "
          f"std {d.std():.4f}, per-image max {np.abs(d).max():.4f}")
    assert np.isfinite(res["bf16"]).all()
    assert abs(d.mean()) < 0.01, d.mean()
    assert np.abs(d).max() < 0.1, np.abs(d).max()
    assert f"{p16.mean():.2f}" == f"{p32.mean():.2f}" or abs(p16.mean() - p32.mean()) < 0.005   # "to 2 decimals" up to a rounding boundary
