"""north_star: "identical PSNR to 2 decimals" -- for the arithmetic the benchmark runs (bf16 operands) at the benchmarked
size.  16 images at 256 x 256, JPEG and WebP q10, the harness' 80-step schedule (webp_inference.py:678-679), device noise:
restored PSNR of bf16 mode against fp32 parity mode (which tracks the reference to < 1e-3 per pixel, test_gpu_model.py).

The sampling loop is chaotic (uint8 truncation + a lossy codec every step), so two arithmetic modes do not
restore the same pixels; what is equal is the PSNR: the per-image difference is a scatter of std 0.013-0.015 dB around
+0.003 dB (tools/psnr256_bias.py, 64 images, profiles/r03_psnr256_bias.txt: bf16 - fp32 = +0.0034 +- 0.0016 dB; with the bf16
activation stream +0.0031 +- 0.0018; single 16-image draws of this test have given +0.001, +0.007 and +0.010).  32 images
(two batches, standard error of the mean 0.0023 dB) keep the bar |mean| < 0.01 dB three standard errors away from that bias;
per image < 0.1 dB.  The mean is a statistic of the set, "equal to 2 decimals" within its standard error -- a printed-digits
comparison would flip at rounding boundaries and is not asserted.
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
    steps = harness.steps_for_quality(arch, q)
    assert steps == 80
    batches = []
    for bi in range(2):
        x0 = synth.synth_images(n_img, R, seed=777 + bi)
        batches.append((x0, harness.COMPRESS[arch](torch.from_numpy(x0), q).cuda()))
    ps = {}
    for mode in ("f32", "bf16"):
        m = MODEL_BY_ARCH[arch](device=torch.device("cuda", 0), dtype=mode)
        m.load_state_dict(synth_state(arch))
        out = [SAMPLER_BY_ARCH[arch](m).sample(y, q, steps=steps, seed=3 + bi).cpu().numpy() for bi, (_, y) in enumerate(batches)]
        assert all(np.isfinite(o).all() for o in out)
        ps[mode] = np.concatenate([psnr_each(o, x0) for o, (x0, _) in zip(out, batches)])
        del m
        torch.cuda.empty_cache()
    p32, p16 = ps["f32"], ps["bf16"]
    d = p16 - p32
    name = {v: k for k, v in spec.ARCH_NAMES.items()}[arch]
    print(f"{name}: restored fp32 {p32.mean():.4f} dB; bf16 - fp32 mean {d.mean():+.5f} dB, "
          f"std {d.std():.4f}, per-image max {np.abs(d).max():.4f}")
    assert abs(d.mean()) < 0.01, d.mean()
    assert np.abs(d).max() < 0.1, np.abs(d).max()
